"""MI355X-native SSL embedding extraction (the hot path of AI-Unicamp/interspeech_ser).

Only the extraction path is here: host-side mirrors of the reference's
``preprocessing/preprocess_speech.py`` / ``preprocess_whisper.py`` drivers on top
of ``libserhip.so``, a C-ABI library of hand-written gfx950 kernels
(include/ser_hip.h).  Importing this package does not load the library; creating
an engine does, and raises if it is absent -- there is no CPU fallback.
"""
__version__ = "0.1.0"

#!/usr/bin/env python
"""Same command line as the reference's preprocessing/preprocess_whisper.py:

    python preprocessing/preprocess_whisper.py --ssl_type openai/whisper-large-v3 --wav_dir W --save_path S
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd.driver import run_whisper  # noqa: E402

if __name__ == "__main__":
    sys.exit(run_whisper())

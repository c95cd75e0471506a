#!/usr/bin/env python
"""Same command line as the reference's preprocessing/preprocess_speech.py:

    python preprocessing/preprocess_speech.py --ssl_type microsoft/wavlm-large --wav_dir W --save_path S
    torchrun --nproc-per-node 8 preprocessing/preprocess_speech.py ...      # one process per MI355X
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd.driver import run_speech  # noqa: E402

if __name__ == "__main__":
    sys.exit(run_speech())

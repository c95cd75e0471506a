#!/usr/bin/env python
"""Same command line as the reference's preprocessing/preprocess_roberta.py:

    python preprocessing/preprocess_roberta.py --roberta_type roberta-large --df_path labels.csv --save_path S
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd.driver import run_roberta  # noqa: E402

if __name__ == "__main__":
    sys.exit(run_roberta())

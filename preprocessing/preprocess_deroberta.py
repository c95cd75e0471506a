#!/usr/bin/env python
"""Same command line as the reference's preprocessing/preprocess_deroberta.py (its flag is called --roberta_type too):

    python preprocessing/preprocess_deroberta.py --roberta_type microsoft/deberta-v3-large --df_path labels.csv --save_path S
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd.driver import run_deberta  # noqa: E402

if __name__ == "__main__":
    sys.exit(run_deberta())

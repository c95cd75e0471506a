#!/usr/bin/env python
"""Counterpart of the reference's bin/eval_cat_bimodal_lazy_1head.py: Development split through ``multimodal_ser.pt``,
macro-F1 and ``results/dev.csv`` (interspeech_ser_amd/head.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd.head import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main(evaluate_only=True))

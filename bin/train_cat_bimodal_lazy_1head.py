#!/usr/bin/env python
"""Counterpart of the reference's bin/train_cat_bimodal_lazy_1head.py: same ``--seed`` / ``--config_path`` flags, same config
keys; trains the bimodal fusion head on the feature files the extraction drivers wrote (interspeech_ser_amd/head.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd.head import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())

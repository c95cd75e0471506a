#!/usr/bin/env python3
"""Embedding extraction with a LoRA fine-tuned Whisper encoder (the reference's
preprocessing/preprocess_whisper_pretrained.py): same command line as preprocess_whisper.py; the fine-tuned
checkpoint -- hard-coded in the reference (:183) -- is passed with --checkpoint (and --lora_alpha if it was not
trained with the reference's LoraConfig(r=8, lora_alpha=16, target_modules=['q_proj', 'v_proj'])).  The adapters
of the encoder are folded into its base weights at load (interspeech_ser_amd.weights.merge_lora; the decoder and
the classifier head of the fine-tuning wrapper are dropped); extraction itself is preprocess_whisper.py's.

    python preprocessing/preprocess_whisper_pretrained.py --ssl_type openai/whisper-large-v3 \
        --checkpoint experiments/LORA_WHISPER_LARGE_V3/whisper_lora_ser.pt --wav_dir W --save_path S
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from interspeech_ser_amd.driver import run_whisper  # noqa: E402

if __name__ == "__main__":
    if not any(a == "--checkpoint" or a.startswith("--checkpoint=") for a in sys.argv[1:]):
        print("Error: --checkpoint <fine-tuned state dict> is required (the reference hard-codes its path)")
        print("Something went wrong, make sure everything is correct before running again!")
        sys.exit(0)
    sys.exit(run_whisper(sys.argv[1:]))

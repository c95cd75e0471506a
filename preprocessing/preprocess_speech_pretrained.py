#!/usr/bin/env python3
"""Embedding extraction with a LoRA fine-tuned WavLM (the reference's
preprocessing/preprocess_speech_pretrained.py): same command line as preprocess_speech.py; the fine-tuned
checkpoint -- hard-coded in the reference (:171) -- is passed with --checkpoint (and --lora_alpha if it was
not trained with the reference's LoraConfig(r=8, lora_alpha=16)).  The adapters are folded into the base
weights at load (interspeech_ser_amd.weights.merge_lora); extraction itself is preprocess_speech.py's.

    python preprocessing/preprocess_speech_pretrained.py --ssl_type microsoft/wavlm-large \
        --checkpoint experiments/LORA_WAVLMLARGE/whisper_lora_ser.pt --wav_dir W --save_path S
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from interspeech_ser_amd.driver import run_speech  # noqa: E402

if __name__ == "__main__":
    if not any(a == "--checkpoint" or a.startswith("--checkpoint=") for a in sys.argv[1:]):
        print("Error: --checkpoint <fine-tuned state dict> is required (the reference hard-codes its path)")
        print("Something went wrong, make sure everything is correct before running again!")
        sys.exit(0)
    sys.exit(run_speech(sys.argv[1:]))

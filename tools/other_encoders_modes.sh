#!/bin/bash
# HuBERT-xlarge / XLS-R-2B / Whisper-large-v3 steps per numerics mode (GPU box): value, verified, whole-path TFLOP/s.
cd "$(dirname "$0")/.."
for cfg in "facebook/hubert-xlarge-ll60k" "facebook/wav2vec2-xls-r-2b --batch 8" "openai/whisper-large-v3 --seconds 30 --reps 4 --steps 3"; do
  for m in bf16 f16a fp32x; do
    python bench.py --other-encoders none --ssl_type $cfg --mode $m --no-cpu-baseline --no-parity --no-e2e --no-trace 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$cfg', '$m', d['value'], d.get('verified'), d.get('achieved_tflops_whole_path'))"
  done
done

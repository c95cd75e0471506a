#!/bin/bash
# A/B of the driver pipeline depth on one box: SER_PIPE_SLOTS=2 (round-2 form: the GPU drops to one batch while the launching thread
# collects / prepares) against 3 (a third batch waits behind an event; at most two compute at once).  bench.py's end-to-end leg.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/pipe_slots_ab.txt
mkdir -p gpurun_out
: > $OUT
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); e=d["end_to_end"]; print(d["value"], "kernel |", e.get("min"), e.get("median"), e.get("max"), "e2e |", e.get("launch_thread_s"), "| rule0", (e.get("reference_default_layer_rule") or {}).get("value"), e.get("error"))'
for rep in 1 2; do
for n in 2 3; do
  echo "== wavlm bf16, SER_PIPE_SLOTS=$n (rep $rep)" | tee -a $OUT
  SER_PIPE_SLOTS=$n python bench.py --other-encoders none --no-cpu-baseline --no-parity --no-trace --steps 10 2>/dev/null | python -c "$pick" | tee -a $OUT
done
done
for n in 2 3; do
  echo "== wavlm f16a, SER_PIPE_SLOTS=$n" | tee -a $OUT
  SER_PIPE_SLOTS=$n python bench.py --other-encoders none --mode f16a --no-cpu-baseline --no-parity --no-trace --steps 10 2>/dev/null | python -c "$pick" | tee -a $OUT
  echo "== whisper bf16, SER_PIPE_SLOTS=$n" | tee -a $OUT
  SER_PIPE_SLOTS=$n python bench.py --other-encoders none --ssl_type openai/whisper-large-v3 --no-cpu-baseline --no-parity --no-trace --steps 5 --e2e-files 1024 2>/dev/null | python -c "$pick" | tee -a $OUT
done

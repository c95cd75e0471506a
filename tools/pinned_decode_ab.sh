#!/bin/bash
# A/B on one box: the driver decodes into page-locked memory and uploads one async copy per utterance (default) against decoding into
# pageable arrays and packing the batch into a pinned staging buffer on the launching thread (SER_PINNED_DECODE=0).  bench.py's end-to-end leg.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/pinned_decode_ab.txt
mkdir -p gpurun_out
: > $OUT
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); e=d["end_to_end"]; print(d["value"], "kernel |", e.get("min"), e.get("median"), e.get("max"), "e2e |", e.get("launch_thread_s"), "| rule0", (e.get("reference_default_layer_rule") or {}).get("value"), e.get("error"))'
for rep in 1 2; do
for n in 0 1; do
  echo "== SER_PINNED_DECODE=$n (rep $rep)" | tee -a $OUT
  SER_PINNED_DECODE=$n python bench.py --other-encoders none --no-cpu-baseline --no-parity --no-trace --steps 10 2>/dev/null | python -c "$pick" | tee -a $OUT
done
done

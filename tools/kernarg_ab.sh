#!/bin/bash
# Where the HIP runtime keeps kernel arguments (HIP_FORCE_DEV_KERNARG: 1 = device memory, 0 = host-coherent memory read over the fabric by
# every block's first scalar loads): the same step with the variable unset / 0 / 1, one box.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/kernarg_ab.txt
mkdir -p gpurun_out
: > $OUT
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"], d.get("verified"), "roofline", d["roofline"].get("frac"), d["roofline"].get("avg_launch_us"))'
for rep in 1 2; do
  echo "== unset (rep $rep)" | tee -a $OUT
  python bench.py --other-encoders none --no-cpu-baseline --no-parity --no-e2e 2>/dev/null | python -c "$pick" | tee -a $OUT
  for v in 0 1; do
    echo "== HIP_FORCE_DEV_KERNARG=$v (rep $rep)" | tee -a $OUT
    HIP_FORCE_DEV_KERNARG=$v python bench.py --other-encoders none --no-cpu-baseline --no-parity --no-e2e 2>/dev/null | python -c "$pick" | tee -a $OUT
  done
done

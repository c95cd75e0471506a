#!/bin/bash
# Same-box A/B of the two forms of ser_attention: the kernel alone, then the bf16 step.   bash tools/attn_res_ab.sh
cd "$(dirname "$0")/.."
OUT=gpurun_out/attn_res_ab.txt
mkdir -p gpurun_out
: > $OUT
for v in 0 1; do
  echo "== SER_ATTN_RESIDENT=$v: kernel alone" | tee -a $OUT
  SER_ATTN_RESIDENT=$v python tools/attn_res_bench.py 2>/dev/null | tee -a $OUT
done
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"], d.get("verified"), d.get("attention_block"))'
for rep in 1 2; do
for v in 0 1; do
  echo "== SER_ATTN_RESIDENT=$v step (rep $rep)" | tee -a $OUT
  SER_ATTN_RESIDENT=$v python bench.py --other-encoders none --no-cpu-baseline --no-parity --no-e2e "$@" 2>/dev/null | python -c "$pick" | tee -a $OUT
done
done

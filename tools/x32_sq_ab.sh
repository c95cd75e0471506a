#!/bin/bash
# (needs the experiments build: make -C interspeech_ser_amd/csrc clean all EXPERIMENTS=1 -- the tile-selection knobs are constants in the product library)
# A/B on one box (env knobs, one build): which two-plane launches take the 256x256 tile (SER_GEMM_X32_SQ_MIN = least number of tiles).
# 150 (default): packed projection only; 100: the output projection too (128 tiles at M = 7 984: half the chip, like FC2's deep-K tile).
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/x32_sq_ab.txt
mkdir -p gpurun_out
: > $OUT
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"], d.get("verified"))'
for rep in 1 2; do
for v in 150 100; do
  echo "== f16a, SER_GEMM_X32_SQ_MIN=$v (rep $rep)" | tee -a $OUT
  SER_GEMM_X32_SQ_MIN=$v python bench.py --other-encoders none --mode f16a --no-cpu-baseline --no-parity --no-e2e --no-trace --steps 10 2>/dev/null | python -c "$pick" | tee -a $OUT
done
done
for v in 150 100; do
  echo "== fp32x, SER_GEMM_X32_SQ_MIN=$v" | tee -a $OUT
  SER_GEMM_X32_SQ_MIN=$v python bench.py --other-encoders none --mode fp32x --no-cpu-baseline --no-parity --no-e2e --no-trace --steps 10 2>/dev/null | python -c "$pick" | tee -a $OUT
done

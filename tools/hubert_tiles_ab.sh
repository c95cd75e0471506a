#!/bin/bash
# (needs the experiments build: make -C interspeech_ser_amd/csrc clean all EXPERIMENTS=1 -- the tile-selection knobs are constants in the product library)
# HuBERT-xlarge (D = 1280: 5 column tiles of 256, 10 of 128) -- tile configurations of the output projection (N = K = 1280) and FC2
# (N = 1280, K = 5120) forced through SER_GEMM_FORCE="N:K:cfg" (0 = 128x128, 1 = 256x128, 2 = 256x256), same box, the step.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/hubert_tiles_ab.txt
mkdir -p gpurun_out
: > $OUT
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"], d.get("verified"))'
for rep in 1 2; do
for f in "" "1280:1280:0" "1280:1280:1" "1280:5120:2" "1280:5120:0"; do
  echo "== SER_GEMM_FORCE='$f' (rep $rep)" | tee -a $OUT
  SER_GEMM_FORCE="$f" python bench.py --other-encoders none --ssl_type facebook/hubert-xlarge-ll60k --no-cpu-baseline --no-parity --no-e2e --no-trace --steps 5 2>/dev/null | python -c "$pick" | tee -a $OUT
done
done

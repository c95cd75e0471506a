#!/bin/bash
# Round 3 experiment: the persistent tile-loop form of ser_gemm (SER_GEMM_PERSIST=1) against the one-block-per-tile form, on the real step.
#   bash tools/gemm_persist_ab.sh        (GPU box; prints two A/B pairs and the integer-exact check under the knob)
F="--no-trace --no-parity --no-e2e --no-cpu-baseline --no-verify --steps 10"
SER_GEMM_PERSIST=1 python -m pytest tests/test_gpu_kernels.py -q -k "integer_exact" 2>&1 | tail -2
for p in 0 1 0 1; do
  SER_GEMM_PERSIST=$p python bench.py $F > gpurun_out/ab_tmp.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('persist=$p', d['value'], 'utt/s')"
done
for p in 0 1; do
  SER_GEMM_PERSIST=$p python bench.py $F --inflight 1 --micro 1 > gpurun_out/ab_tmp.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('one batch alone, persist=$p', d['value'], 'utt/s')"
done

#!/bin/bash
# Round 3 experiment: the persistent (item-walking) form of ser_attention (SER_ATTN_PERSIST=1: 512 resident blocks) against one block per
# (utterance, head, q-tile), in isolation at 16 x 499 frames and on the real step.     bash tools/attn_persist_ab.sh   (GPU box)
F="--no-trace --no-parity --no-e2e --no-cpu-baseline --no-verify --steps 10"
SER_ATTN_PERSIST=1 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -k "attention or golden_ragged or batched_equals" 2>&1 | tail -2
for p in 0 1; do echo "isolated, persist=$p"; SER_ATTN_PERSIST=$p python tools/attn_modes_bench.py 2>&1 | grep "BF16" | grep "bias=1"; done
for p in 0 1 0 1; do
  SER_ATTN_PERSIST=$p python bench.py $F > gpurun_out/ab_tmp.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('step, persist=$p', d['value'], 'utt/s')"
done

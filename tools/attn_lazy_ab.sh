#!/bin/bash
# A/B of the attention softmax forms on one box: HEAD build / packed sub+add (SER_ATTN_LAZY=0) / stale running maximum (SER_ATTN_LAZY=1).
# Libraries are built beforehand as interspeech_ser_amd/lib/libserhip_{head,l0,l1}.so.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/attn_lazy_ab.txt
mkdir -p gpurun_out
: > $OUT
for v in l1 head; do
  echo "== $v: attention tests" | tee -a $OUT
  SER_HIP_LIB=$PWD/interspeech_ser_amd/lib/libserhip_$v.so python -m pytest tests/test_gpu_kernels.py tests/test_gpu_f16q.py -q -k "attention" 2>&1 | tail -15 | tee -a $OUT
done
for v in head l0 l1; do
  echo "== $v: attention per mode" | tee -a $OUT
  SER_HIP_LIB=$PWD/interspeech_ser_amd/lib/libserhip_$v.so python tools/attn_modes_bench.py 2>&1 | tee -a $OUT
done
for rep in 1 2; do
for v in head l0 l1; do
  echo "== $v: whisper step (rep $rep)" | tee -a $OUT
  SER_HIP_LIB=$PWD/interspeech_ser_amd/lib/libserhip_$v.so python bench.py --other-encoders none --ssl_type openai/whisper-large-v3 --no-cpu-baseline --no-parity --no-e2e --no-trace --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d.get('verified'))" | tee -a $OUT
  echo "== $v: wavlm step (rep $rep)" | tee -a $OUT
  SER_HIP_LIB=$PWD/interspeech_ser_amd/lib/libserhip_$v.so python bench.py --other-encoders none --no-cpu-baseline --no-parity --no-e2e --no-trace 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d.get('verified'))" | tee -a $OUT
done
done

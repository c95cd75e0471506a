#!/bin/bash
# Generic same-box A/B of two builds of libserhip on the step: interspeech_ser_amd/lib/libserhip_head.so (previous commit, built by hand)
# against the in-tree library.   bash tools/lib_ab.sh [bench.py args...]
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/lib_ab.txt
mkdir -p gpurun_out
: > $OUT
LIBD=$PWD/interspeech_ser_amd/lib
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"], d.get("verified"))'
for rep in 1 2 3; do
for v in head new; do
  L=$LIBD/libserhip_$v.so; [ $v = new ] && L=$LIBD/libserhip.so
  echo "== $v (rep $rep) $*" | tee -a $OUT
  SER_HIP_LIB=$L python bench.py --other-encoders none --no-cpu-baseline --no-parity --no-e2e --no-trace "$@" 2>/dev/null | python -c "$pick" | tee -a $OUT
done
done

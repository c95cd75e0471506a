#!/bin/bash
# A/B on one box: WavLM gate pre-activations as 2H extra columns of the packed projection (SER_GATE_IN_ATTN=0, rounds 1-3) against
# computed inside ser_attention from the layer input's operand copy (default).  Step of the headline + the f16a parity mode.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/gate_in_attn_ab.txt
mkdir -p gpurun_out
: > $OUT
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["ms_per_step"], d.get("verified"), d["verification"].get("timed_mode_max_rel_err_vs_oracle"), (d.get("attention_block") or {}).get("frac"))'
for rep in 1 2; do
for v in 0 1; do
  echo "== bf16, SER_GATE_IN_ATTN=$v (rep $rep)" | tee -a $OUT
  SER_GATE_IN_ATTN=$v python bench.py --other-encoders none --no-cpu-baseline --no-parity --no-e2e --no-trace 2>/dev/null | python -c "$pick" | tee -a $OUT
  echo "== f16a, SER_GATE_IN_ATTN=$v (rep $rep)" | tee -a $OUT
  SER_GATE_IN_ATTN=$v python bench.py --other-encoders none --mode f16a --no-cpu-baseline --no-parity --no-e2e --no-trace --steps 10 2>/dev/null | python -c "$pick" | tee -a $OUT
done
done

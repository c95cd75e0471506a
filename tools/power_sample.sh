#!/bin/bash
# Board power / clocks while the timed step runs (tools/, GPU box): bash tools/power_sample.sh [bench.py args...]
# Is the step power-bound?  (MI355X_MICROARCH.md "DVFS give-back": an in-kernel cycle saving can return as a lower clock instead of throughput.)
cd "$(dirname "$0")/.."
OUT=gpurun_out/power_sample.txt
mkdir -p gpurun_out; : > $OUT
( for i in $(seq 1 60); do echo "--- t=$i" >> $OUT; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction)" >> $OUT; sleep 1; done ) &
SP=$!
python bench.py --no-cpu-baseline --no-e2e --no-trace --no-parity --other-encoders none --steps 200 "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("bench:", d["value"], d["ms_per_step"], d.get("verified"))' | tee -a $OUT
kill $SP 2>/dev/null
wait $SP 2>/dev/null
echo done >> $OUT

#!/bin/bash
# bash tools/power_beside.sh <command ...>   (GPU box): runs the command, samples rocm-smi every 0.3 s beside it and reports power / sclk for every output line of the form
#   "<name> first launch ... wall <t0> .. <t1> .. <t2>"  (tools/energy_probe.hip, tools/gemm_power.py)
# rocm-smi power / clock samples with wall-clock stamps beside the probe's own stamps -> gpurun_out/energy_probe.txt
cd "$(dirname "$0")/.."
OUT=gpurun_out/power_beside.txt
mkdir -p gpurun_out; : > $OUT; : > $OUT.smi
( for i in $(seq 1 400); do echo "--- t=$(date +%s.%N)" >> $OUT.smi; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" >> $OUT.smi; sleep 0.3; done ) &
SP=$!
timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids | tee -a $OUT
kill $SP 2>/dev/null
wait $SP 2>/dev/null
python3 - "$OUT" <<'PY'
import re, sys
out = sys.argv[1]
samples, t = [], None
for line in open(out + ".smi"):
    m = re.match(r"--- t=([\d.]+)", line)
    if m:
        t = float(m.group(1)); continue
    m = re.search(r"Power \(W\): ([\d.]+)", line) or re.search(r"Socket Power.*?: ([\d.]+)", line)
    if m and t:
        samples.append([t, float(m.group(1)), None])
    m = re.search(r"sclk.*\((\d+)Mhz\)", line)
    if m and samples and samples[-1][2] is None:
        samples[-1][2] = int(m.group(1))
rows = []
for line in open(out):
    m = re.match(r"(\S+)\s+first launch.*wall ([\d.]+) \.\. ([\d.]+) \.\. ([\d.]+)", line)
    if m:
        name, t0, t1, t2 = m.group(1), float(m.group(2)), float(m.group(3)), float(m.group(4))
        s = [x for x in samples if t0 + 0.8 < x[0] < t2]
        if s:
            rows.append(f"{name:12s} power {min(x[1] for x in s):6.0f} .. {max(x[1] for x in s):6.0f} W (mean {sum(x[1] for x in s) / len(s):6.0f}, {len(s)} samples)   sclk {min(x[2] or 0 for x in s)} .. {max(x[2] or 0 for x in s)} MHz")
open(out, "a").write("# rocm-smi beside each variant (from 0.8 s after its start):\n" + "\n".join(rows) + "\n")
print("\n".join(rows))
PY

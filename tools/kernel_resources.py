"""Register / scratch use of the kernels in a hipcc -S listing (tools/, build-time check): python tools/kernel_resources.py file.s [substring]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
pat = re.compile(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?"
                 r"\s+\.vgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)")
for m in pat.finditer(s):
    dn = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if want in dn:
        print(f"{dn[:150]:150s} scratch {m.group(2):>4} sgpr {m.group(3):>3} vgpr {m.group(4):>3} spill {m.group(5)}")

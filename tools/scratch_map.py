"""Where a kernel's scratch (spill) accesses sit relative to its MFMA blocks (tools/, build-time check):
python tools/scratch_map.py file.s <mangled-name substring>"""
import re
import sys

s = open(sys.argv[1]).read()
idx = [m.start() for m in re.finditer(r"^_Z\w+: +; @", s, re.M)]
for i, st in enumerate(idx):
    f = s[st: idx[i + 1] if i + 1 < len(idx) else len(s)]
    name = f.split(":")[0]
    if sys.argv[2] not in name:
        continue
    blocks, cur, lab = [], [], "entry"
    for line in f.split("\n"):
        m = re.match(r"(\.LBB\d+_\d+):", line)
        if m:
            blocks.append((lab, cur))
            cur, lab = [], m.group(1)
        else:
            cur.append(line)
    blocks.append((lab, cur))
    print(name)
    for lab, b in blocks:
        nm, ns = sum("v_mfma" in x for x in b), sum("scratch_" in x for x in b)
        if nm or ns:
            print(f"   {lab:12s} mfma {nm:4d} scratch {ns:4d} instructions {sum(1 for x in b if x.startswith(chr(9)) and not x.strip().startswith(';'))}")

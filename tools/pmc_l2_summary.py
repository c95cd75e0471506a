"""Per-kernel L2 hit rate from one rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum pass (MI355X guide, "L2 (per XCD)"):
hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), summed over the launches of a kernel.  It is what separates
"bytes through the fabric" (FETCH_SIZE: every XCD pulls its own copy of a shared operand panel through its own 4 MiB L2;
Infinity-Cache hits are counted) from wasted re-reads inside an XCD.

    python tools/pmc_l2_summary.py gpurun_out/prof/pmc_l2 profiles/r02
"""
import collections, csv, glob, json, sys
f = glob.glob(f"{sys.argv[1]}/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    n[r["Kernel_Name"]].add(r["Dispatch_Id"])
rows = []
for k, c in sorted(acc.items(), key=lambda kv: -(kv[1].get("TCC_HIT_sum", 0) + kv[1].get("TCC_MISS_sum", 0))):
    if not ("ser_gemm" in k or "attention" in k or "logmel" in k or "layernorm" in k or "row_center" in k or "wave_" in k):
        continue
    h, m = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    rows.append({"kernel": k, "launches": len(n[k]), "l2_hit_rate": round(h / max(h + m, 1.0), 4),
                 "l2_requests_per_launch": round((h + m) / max(len(n[k]), 1)), "l2_misses_per_launch": round(m / max(len(n[k]), 1))})
json.dump({"command": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -- python3 bench.py (eager, tools/profile_all.sh)",
           "kernels": rows}, open(f"{sys.argv[2]}_pmc_l2.json", "w"), indent=1)
for r in rows:
    print(r["kernel"][:80], r["l2_hit_rate"], r["l2_requests_per_launch"], r["l2_misses_per_launch"])

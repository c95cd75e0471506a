"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as the MI355X guide
prescribes) into profiles/<tag>_pmc_traffic.json + a per-kernel CSV.

FETCH_SIZE / WRITE_SIZE are reported in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE reads exactly half of the bytes of a wide coalesced (16 B/lane) streaming read, so it is
doubled; WRITE_SIZE is exact for 16-B-per-lane stores.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02 "<workload key of bench.py>"

The summary is stamped with the digest of the kernel sources (bench.kernel_source_digest) and the workload key, so that
bench.py quotes `roofline.traffic` only for the kernels and the workload the counters were taken from.
"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def load(d, counter):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return out

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
tag = sys.argv[3]
rows, gemm_f, gemm_w, gemm_n = [], 0.0, 0.0, 0
for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
    n = len(fetch[k])
    f_raw = sum(fetch[k]) / n
    w = sum(write.get(k, [0.0])) / max(1, len(write.get(k, [])))
    rows.append({"kernel": k, "launches": n, "fetch_raw_bytes_per_launch": round(f_raw),
                 "fetch_corrected_bytes_per_launch": round(2 * f_raw), "write_bytes_per_launch": round(w)})
    if "ser_gemm_kernel" in k:
        gemm_f += 2 * sum(fetch[k]); gemm_w += sum(write.get(k, [0.0])); gemm_n += n
with open(f"{tag}_pmc_per_kernel.csv", "w", newline="") as fh:
    wcsv = csv.DictWriter(fh, fieldnames=list(rows[0]))
    wcsv.writeheader(); wcsv.writerows(rows)
import bench
summary = {"kernel": "ser_gemm_kernel (all tile configs)", "launches": gemm_n,
           "kernel_source_digest": bench.kernel_source_digest(), "workload_key": sys.argv[4] if len(sys.argv) > 4 else "",
           "hbm_bytes_per_launch": round((gemm_f + gemm_w) / gemm_n),
           "fetch_corrected_bytes_per_launch": round(gemm_f / gemm_n), "write_bytes_per_launch": round(gemm_w / gemm_n),
           "correction": "FETCH_SIZE x2 (gfx950, 16 B/lane streaming reads), WRITE_SIZE x1; KiB -> bytes",
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py "
                      "--steps 2 --warmup 1 --reps 1 --no-graph --no-verify --no-trace --no-cpu-baseline --no-parity --no-e2e "
                      "(two separate passes, tools/profile_all.sh)"}
json.dump(summary, open(f"{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(summary))

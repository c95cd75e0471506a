"""Summarise one rocprofv3 --pmc pass of SQ counters per kernel into profiles/<tag>_pmc_sq.json.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \\
              SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- \\
              python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-trace --no-graph
    python tools/pmc_sq_summary.py DIR profiles/r01

MfmaUtil = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 1024 SIMDs).  rocprofv3 reports GRBM_GUI_ACTIVE summed over
the 8 XCDs (checked: per launch it is 8 x duration x clock) and SQ_VALU_MFMA_BUSY_CYCLES summed over all SIMDs (checked:
it equals algorithmic FLOPs / 1024 for v_mfma_f32_16x16x32_bf16, 16 busy cycles per 16 384 FLOPs), so kernel cycles =
GRBM_GUI_ACTIVE / 8."""
import collections, csv, glob, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
dirs = [sys.argv[1]] + sys.argv[3:]               # optional further passes (other counters of the same command)
for di, d in enumerate(dirs):
    fs = glob.glob(f"{d}/*/*counter_collection.csv")
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        name = r["Counter_Name"]
        if di > 0 and name in ("GRBM_GUI_ACTIVE", "SQ_BUSY_CU_CYCLES"):
            name += f"_pass{di + 1}"                # each pass is normalised by its own cycle counts
        acc[k][name] += float(r["Counter_Value"])
        if di == 0:
            launches[k].add(r["Dispatch_Id"])
rows = []
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    if not ("ser_gemm" in k or "attention" in k):
        continue
    gui = max(c.get("GRBM_GUI_ACTIVE", 0.0), 1.0)
    rows.append({
        "kernel": k, "launches": len(launches[k]),
        "mfma_util_pct": round(100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024), 1),
        "lds_bank_conflict_pct_of_lds_active": round(100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 2),
        "wave_parked_pct (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": round(100.0 * c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0), 1),
        "issue_stall_pct (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": round(100.0 * c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0), 1),
        # LDS array busy share of the CU-busy time (SQ_LDS_IDX_ACTIVE counts LDS-array cycles per CU; 256 B/clk/CU when active)
        "lds_active_pct_of_cu_busy (SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES)": round(100.0 * c.get("SQ_LDS_IDX_ACTIVE", 0.0) / max(c.get("SQ_BUSY_CU_CYCLES", 0.0), 1.0), 1),
        "valu_mfma_coexec_pct_of_mfma_busy": (round(100.0 * c["SQ_VALU_MFMA_COEXEC_CYCLES"] / max(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) *
                                              c.get("GRBM_GUI_ACTIVE_pass2", gui) / gui, 1.0), 1) if "SQ_VALU_MFMA_COEXEC_CYCLES" in c else None),
        "raw": {n: int(v) for n, v in c.items()},
    })
json.dump({"command": "see tools/pmc_sq_summary.py", "kernels": rows}, open(f"{sys.argv[2]}_pmc_sq.json", "w"), indent=1)
for r in rows:
    print(r["kernel"][:70], {k: v for k, v in r.items() if k not in ("kernel", "raw")})

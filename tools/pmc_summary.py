"""Per-launch HBM bytes and SQ ratios of the MFMA kernel classes from the separate rocprofv3 --pmc passes of tools/prof_all.sh.
usage: python tools/pmc_summary.py gpurun_out [bf16|f16] > profiles/r2_pmc_bf16.json
FETCH_SIZE is doubled (gfx950: wide coalesced reads are tallied at 1/2, MI355X_MICROARCH.md, HBM section); units KB.
The output records `csrc_sha` = the hash of floodplanet_code_amd/csrc/*.{hip,h} the profiled library was built from (the box
runs a snapshot of this tree): bench.py reports `roofline.traffic` from this file only while its own sources hash the same."""
import csv, glob, json, os, sys, statistics as st
root = sys.argv[1]
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# class key (what fu_profile_read / bench.py call the class) -> substring of the kernel names it covers
KERNELS = {f"k_conv3x3_{dt}": f"k_conv3x3_{dt}", f"k_wgrad_{dt}": f"k_wgrad_{dt}"}

def rows(sub):
    out = []
    import os
    fns = sorted(glob.glob(f"{root}/{sub}/*/*_counter_collection.csv"), key=os.path.getmtime)
    if fns:   # gpurun_out accumulates runs: newest only
        out = list(csv.DictReader(open(fns[-1])))
    return out

def per_kernel(sub, counter):
    acc = {}
    for r in rows(sub):
        if r["Counter_Name"] != counter:
            continue
        for key, pat in KERNELS.items():
            if pat in r["Kernel_Name"]:
                acc.setdefault(key, {}).setdefault(r["Dispatch_Id"], 0.0)
                acc[key][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return acc

fetch, write = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
sq = {c: per_kernel("pmc_sq", c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")}
out = {}
for k in KERNELS:
    f = list(fetch.get(k, {}).values()); w = list(write.get(k, {}).values())
    if not f or not w:
        continue
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    d = {"launches_profiled": len(f), "avg_FETCH_SIZE_KB": round(fk, 1), "avg_WRITE_SIZE_KB": round(wk, 1),
         "hbm_bytes_per_launch": int((2 * fk + wk) * 1024),
         "note": "separate --pmc passes; gfx950 FETCH_SIZE doubled (wide coalesced reads are tallied at 1/2, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is"}
    ids = sq["SQ_BUSY_CU_CYCLES"].get(k, {})
    if ids:
        mf = [sq["SQ_VALU_MFMA_BUSY_CYCLES"][k][i] / (4 * sq["SQ_BUSY_CU_CYCLES"][k][i]) for i in ids if sq["SQ_BUSY_CU_CYCLES"][k][i] > 0]
        la = [sq["SQ_LDS_IDX_ACTIVE"][k][i] / sq["SQ_BUSY_CU_CYCLES"][k][i] for i in ids if sq["SQ_BUSY_CU_CYCLES"][k][i] > 0]
        lc = [sq["SQ_LDS_BANK_CONFLICT"][k][i] / sq["SQ_LDS_IDX_ACTIVE"][k][i] for i in ids if sq["SQ_LDS_IDX_ACTIVE"][k][i] > 0]
        d["sq_pmc"] = {"mfma_busy_over_4x_cu_busy_median": round(st.median(mf), 3), "lds_active_over_cu_busy_median": round(st.median(la), 3),
                       "lds_bank_conflict_over_lds_active_median": round(st.median(lc), 3)}
    out[k] = d
import bench  # noqa: E402  (csrc_sha only; nothing touches the GPU)
out["csrc_sha"] = bench.csrc_sha()
out["command"] = "tools/prof_all.sh: rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ set> --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (bf16, B=16, 8ch, 256x256); summarised by tools/pmc_summary.py"
print(json.dumps(out, indent=1))

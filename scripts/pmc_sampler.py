#!/usr/bin/env python3
"""Digest of scripts/pmc_sampler.sh: per kernel of the reverse sampler, per launch: HBM-side bytes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
KiB -> bytes), MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES over all SIMD-cycles of the dispatch), LDS bank-conflict share, wait shares."""
import collections, csv, glob, json, os, re, sys
out, mode = sys.argv[1], sys.argv[2]


def rows(kind):
    f = max(glob.glob(f"{out}/{kind}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    return list(csv.DictReader(open(f)))


def short(k):
    k = re.sub(r"\(.*", "", k).replace("void ", "").replace("mdm::", "")
    return k


def collect(kind):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in rows(kind):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return acc, {k: len(v) for k, v in n.items()}


f, nf = collect("fetch"); w, nw = collect("write"); q, nq = collect("sq")
res = {"command": f"rocprofv3 --kernel-trace --pmc <counters> -- python3 scripts/sampler_time.py {mode} 20 (one run per counter set)",
       "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported; KiB -> bytes x1024",
       "mfma_busy_formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs), summed over the kernel's dispatches",
       "kernels": {}}
for k in sorted(q, key=lambda k: -q[k]["GRBM_GUI_ACTIVE"]):
    s = q[k]
    gui = s["GRBM_GUI_ACTIVE"] / 8.0
    if gui <= 0 or not any(x in k for x in ("conv_halo", "lin_split", "gemm_f32", "gn_fwd", "skinny")):
        continue
    d = {"launches": nq[k]}
    if k in f and k in w:
        d["hbm_bytes_per_launch"] = (2.0 * f[k]["FETCH_SIZE"] / nf[k] + w[k]["WRITE_SIZE"] / nw[k]) * 1024.0
    d["mfma_busy"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 256 * 4)
    d["cu_busy"] = s["SQ_BUSY_CU_CYCLES"] / (gui * 256) if s["SQ_BUSY_CU_CYCLES"] else None
    if s["SQ_WAVE_CYCLES"] > 0:
        d["wait_any_share_of_wave_cycles"] = s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"]
        d["wait_inst_lds_share_of_wave_cycles"] = s["SQ_WAIT_INST_LDS"] / s["SQ_WAVE_CYCLES"]
    if s["SQ_LDS_IDX_ACTIVE"] > 0:
        d["lds_bank_conflict_share_of_lds_cycles"] = s["SQ_LDS_BANK_CONFLICT"] / s["SQ_LDS_IDX_ACTIVE"]
    res["kernels"][k] = d
json.dump(res, open(f"{out}/pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])

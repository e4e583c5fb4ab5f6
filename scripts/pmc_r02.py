#!/usr/bin/env python3
"""Digest of scripts/pmc_r02.sh: per kernel family HBM traffic (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KiB), MFMA busy fraction,
LDS bank-conflict share, wait shares, L2 hit rate.  Family "contraction" = everything an mdm_gemm / mdm_wgrad_group_launch
call launches in the bf16 step (the kernels bench.py's `roofline` times)."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]


def rows(kind):
    f = max(glob.glob(f"{out}/{kind}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    return list(csv.DictReader(open(f)))


def family(k):
    if any(x in k for x in ("conv_halo", "conv_small", "conv_lin2", "conv_pair", "conv_thin", "chain_kernel", "wgrad_group", "wgrad_taps", "tile_parts_reduce",
                            "wgrad_lin", "gemm_ring", "gemm_bf16", "splitk_")):
        return "contraction"
    if "attn_" in k: return "attention"
    if "gn_" in k: return "groupnorm"
    if any(x in k for x in ("adamw", "sqnorm", "transpose_shadow")): return "optimizer"
    return "other"


def collect(kind):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    steps = 0
    seen = set()
    for r in rows(kind):
        k = r["Kernel_Name"]
        acc[family(k)][r["Counter_Name"]] += float(r["Counter_Value"])
        if "adamw_kernel" in k and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); steps += 1
    return acc, steps


res = {"command": "rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sampler (one run per counter set)",
       "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported; KiB -> bytes x1024",
       "mfma_busy_formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs), summed over the family's dispatches",
       "families": {}}
f, nf = collect("fetch"); w, nw = collect("write"); q, nq = collect("sq"); c, nc = collect("tcc")
calls = json.load(open(f"{out}/fetch.json"))["roofline"]["launches_per_step"]
for fam in sorted(set(f) | set(w) | set(q)):
    d = {}
    d["hbm_bytes_per_step"] = (2.0 * f[fam]["FETCH_SIZE"] / nf + w[fam]["WRITE_SIZE"] / nw) * 1024.0
    d["fetch_bytes_per_step"] = 2.0 * f[fam]["FETCH_SIZE"] / nf * 1024.0
    d["write_bytes_per_step"] = w[fam]["WRITE_SIZE"] / nw * 1024.0
    s = q[fam]
    gui = s["GRBM_GUI_ACTIVE"] / 8.0
    if gui > 0:
        d["mfma_busy"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 256 * 4)
        d["cu_busy"] = s["SQ_BUSY_CU_CYCLES"] / (gui * 256) if s["SQ_BUSY_CU_CYCLES"] else None
    if s["SQ_WAVE_CYCLES"] > 0:
        d["wait_any_share_of_wave_cycles"] = s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"]
        d["wait_inst_lds_share_of_wave_cycles"] = s["SQ_WAIT_INST_LDS"] / s["SQ_WAVE_CYCLES"]
    if s["SQ_LDS_IDX_ACTIVE"] > 0:
        d["lds_bank_conflict_share_of_lds_cycles"] = s["SQ_LDS_BANK_CONFLICT"] / s["SQ_LDS_IDX_ACTIVE"]
    t = c[fam]
    if t["TCC_HIT_sum"] + t["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"])
    res["families"][fam] = d
con = res["families"]["contraction"]
res["kernel_family"] = "conv_halo / conv_lin2 / conv_pair / wgrad_group / gemm_ring / gemm_bf16 + splitk_* (all bf16 mdm_gemm, mdm_gemm_pair and mdm_wgrad_group_launch calls)"
res["launches_per_step"] = calls
res["hbm_bytes_per_launch"] = con["hbm_bytes_per_step"] / calls
res["mfma_busy"] = con.get("mfma_busy")
res["steps_counted"] = [nf, nw, nq, nc]
res["git_head"] = os.environ.get("GIT_HEAD")
# Calibration (scripts/pmc_r04.sh): the same counter and formula on scripts/micro/mfma_rate.hip, whose waves issue v_mfma_f32_16x16x32_bf16
# back to back (16.4 cycles per MFMA per SIMD by s_memtime, 2 130 TFLOP/s: the pipe IS busy every cycle).  What the formula reads there
# is the counter's value for "100 % busy"; family busy / that value = share of the matrix pipe's issue slots in use.
try:
    cal = rows("cal")
    a = collections.defaultdict(float)
    for r in cal:
        a[r["Counter_Name"]] += float(r["Counter_Value"])
    full = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
    res["mfma_busy_calibration"] = {"busy_reading_of_a_full_rate_mfma_loop": full, "family_busy_calibrated": (con.get("mfma_busy") or 0.0) / full,
                                    "source": "scripts/micro/mfma_rate.hip under the same counter / formula",
                                    "micro_output": open(f"{out}/cal.txt").read().strip().splitlines()[-2:]}
except Exception as e:      # noqa: BLE001
    res["mfma_busy_calibration"] = {"error": f"{type(e).__name__}: {e}"[:200]}
json.dump(res, open(f"{out}/pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))

#!/usr/bin/env python3
"""Per-category summary of a rocprofv3 kernel_stats.csv: python scripts/prof_summary.py stats.csv <steps profiled>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# divisor = optimizer steps profiled = calls of adamw_kernel (one per step, warm-up and event-timed replays included)
steps = float(sys.argv[2]) if len(sys.argv) > 2 else float(next(int(r["Calls"]) for r in rows if "adamw_kernel" in r["Name"]))
cats = collections.OrderedDict((k, [0, 0.0]) for k in ("wgrad_lin", "conv_pair", "conv_halo", "wgrad ring<.,.,2", "conv_lin", "gemm_ring other", "gemm_bf16", "gemm_f32", "splitk_reduce", "splitk_epilogue", "gn_bwd", "gn_fwd", "softmax", "adamw/sqnorm/transpose", "other"))
def cat(n):
    if "gemm_ring_kernel" in n: return "wgrad ring<.,.,2" if ", 2, " in n.split("(")[0] else "gemm_ring other"
    if "conv_pair" in n: return "conv_pair"
    if "conv_lin" in n: return "conv_lin"
    if "wgrad_lin" in n or "wgrad_group" in n or "wgrad_taps" in n or "tile_parts_reduce" in n: return "wgrad_lin"
    if "conv_halo" in n: return "conv_halo"
    for k in ("gemm_bf16", "gemm_f32", "splitk_reduce", "splitk_epilogue", "gn_bwd", "gn_fwd", "softmax"):
        if k in n: return k
    if any(k in n for k in ("adamw", "sqnorm", "transpose_shadow")): return "adamw/sqnorm/transpose"
    return "other"
tot = 0.0; nk = 0
for r in rows:
    c = cat(r["Name"]); cats[c][0] += int(r["Calls"]); cats[c][1] += float(r["TotalDurationNs"]); tot += float(r["TotalDurationNs"]); nk += int(r["Calls"])
print(f"kernels/step {nk/steps:.0f}, kernel time/step {tot/steps/1e6:.3f} ms")
for k, (n, t) in cats.items():
    print(f"  {k:26s} {n/steps:6.1f} launches/step  {t/steps/1e6:7.3f} ms/step  avg {t/max(n,1)/1e3:6.1f} us")
for r in rows[:18]:
    print(f"    {r['Name'][:84]:84s} {int(r['Calls'])/steps:6.1f}  {float(r['TotalDurationNs'])/steps/1e6:6.3f} ms  {float(r['AverageNs'])/1e3:6.1f} us")

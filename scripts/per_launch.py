#!/usr/bin/env python3
"""Per-launch times of the benchmarked step's forward and backward launch lists (HIP event pair around EVERY C-ABI call,
eager replay, median of 5): python scripts/per_launch.py [--batch 32] > gpurun_out/per_launch.txt
One line per call: list, index, entry point, microseconds, and for contractions the shape / conv geometry / fused flags."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import torch  # noqa: E402

import mdm  # noqa: E402
from mdm import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--products", default="exact", choices=["exact", "split"], help="fp32 only: UNet(f32_products=...)")
ap.add_argument("--fwd-only", action="store_true", help="the forward launch list only (the reverse sampler's U-Net call)")
o = ap.parse_args()
dt = mdm.BF16 if o.dtype == "bf16" else mdm.F32
model = mdm.UNet(mdm.unet6_config(32), N=o.batch, H=32, W=32, dtype=dt, seed=0, use_graph=False, f32_products=o.products)
st = torch.cuda.current_stream().cuda_stream


def describe(name, args):
    if name == "mdm_gemm_pair":
        return describe("mdm_gemm", args[:1]) + "  ||  " + describe("mdm_gemm", args[1:2])
    if name != "mdm_gemm":
        return ""
    d = args[0]._obj            # the call holds C.byref(desc)
    s = f"dt{d.dtype} L{d.layout} M{d.M} N{d.N} K{d.K} b{d.batch}"
    if d.conv:
        s += f" conv {d.KH}x{d.KW} s{d.stride} {d.IH}x{d.IW}->{d.OH}x{d.OW} C{d.C0}+{d.C1} T{d.transposed} ups{d.ups}"
    s += f" sk{d.splitk}"
    for f in ("gnb_x", "gnf_out", "resid", "rowvec", "dbias", "gnb_add", "B_split", "f32_split"):
        if getattr(d, f):
            s += " " + f
    return s


for tag, rec in (("fwd", model.forward_plan),) if o.fwd_only else (("fwd", model.forward_plan), ("bwd", model.backward_plan)):
    runs = []
    for _ in range(5):
        model.zero_grad()
        if tag == "bwd":
            model.forward_plan.run(st)
        runs.append(dict(rec.run_timed(st, lambda i, n: True)))
        torch.cuda.synchronize()
    tot = 0.0
    for i, (name, fn, args) in enumerate(rec.calls):
        if i not in runs[0]:
            continue
        us = 1e3 * sorted(r[i] for r in runs)[2]
        tot += us
        print(f"{tag} {i:4d} {name:28s} {us:8.1f} us  {describe(name, args)}")
    print(f"{tag} total {tot:.1f} us over {len(rec.calls)} calls (event-pair overhead ~2.6 us per call included)")

#!/usr/bin/env python3
"""Per-launch time of single 3x3 convolutions on the sampler's / the train step's layers (HIP events over `reps` back-to-back launches):
    python scripts/conv_time.py [f32_split|f32|bf16] [N=100] [reps=50]
Select another build of the library with MDM_LIB_PATH (A/B of tile variants on one box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import torch
from mdm import _lib, ops
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "f32_split"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dt = 1 if mode == "bf16" else 0
tdt = torch.bfloat16 if mode == "bf16" else torch.float32
LAYERS = [(128, 0, 128, 32, 0), (128, 128, 128, 32, 0), (256, 0, 256, 16, 0), (256, 256, 256, 16, 0), (256, 0, 256, 16, 1), (256, 0, 256, 8, 0), (256, 256, 256, 8, 0),
          (256, 0, 256, 4, 0), (256, 256, 256, 4, 0), (256, 0, 256, 4, 1)]      # unet6 at 32x32: 128 / 256 / 256 / 256 channels
tot = 0.0
for (c0, c1, co, H, ups) in LAYERS:
    g = ops.ConvGeom(N=N, IH=H, IW=H, C0=c0, C1=c1, Cout=co, ups=ups)
    HO = H << ups
    x0 = torch.randn(N, H, H, c0, device=dev).to(tdt)
    x1 = torch.randn(N, H, H, c1, device=dev).to(tdt) if c1 else None
    w = (torch.randn(9, co, c0 + c1, device=dev) * 0.02).to(tdt)
    ws = None
    if mode == "f32_split":
        ws = torch.empty_like(w); segs = torch.tensor([[0, w.numel()]], dtype=torch.int64, device=dev)
        _lib.call("mdm_split_shadow", _lib.ptr(w), _lib.ptr(ws), _lib.ptr(segs), 1, _lib.stream())
    b = torch.zeros(co, device=dev); y = torch.empty(N, HO, HO, co, device=dev, dtype=tdt)
    for _ in range(3):
        ops.conv_fwd(dt, g, x0, x1, w, b, y, w_split=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_fwd(dt, g, x0, x1, w, b, y, w_split=ws)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * N * HO * HO * 9 * (c0 + c1) * co
    tot += us
    print(f"{mode} {c0}+{c1}->{co}@{H}{'^' if ups else ''} N={N}: {us:8.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s", flush=True)
print(f"{mode} sum {tot:.1f} us")

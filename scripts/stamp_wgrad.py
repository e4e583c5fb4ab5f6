#!/usr/bin/env python3
"""Prologue / slab loop / epilogue cycles of wgrad_lin_kernel (library built with `make EXTRA=-DMDM_STAMP`)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import torch, numpy as np
from mdm import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
fn = lib.mdm_debug_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
def stamps(reset=1):
    buf = (ctypes.c_ulonglong * (4096 * 32))(); assert fn(buf, reset) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 32).astype(np.float64)
    return a[a[:, 5] > 0]
for (c0, c1, co, H, k) in [(128, 0, 128, 32, 3), (256, 0, 256, 16, 3), (256, 0, 256, 8, 3), (256, 0, 256, 4, 3), (128, 128, 128, 32, 1)]:
    p = (k - 1) // 2
    g = ops.ConvGeom(N=32, IH=H, IW=H, C0=c0, C1=c1, Cout=co, KH=k, KW=k, stride=1, pad_t=p, pad_l=p, pad_b=p, pad_r=p, ups=0)
    bf = torch.bfloat16
    x0 = torch.randn(32, H, H, c0, device=dev, dtype=bf); x1 = torch.randn(32, H, H, c1, device=dev, dtype=bf) if c1 else None
    dy = torch.randn(32, H, H, co, device=dev, dtype=bf)
    gw = torch.zeros(k * k, co, c0 + c1, device=dev)
    ws = torch.empty(ops.conv_wgrad_ws_bytes(1, g) // 4 + 64, device=dev)
    for _ in range(3): ops.conv_wgrad(1, g, dy, x0, x1, gw, ws=ws)
    torch.cuda.synchronize(); stamps(1)
    ops.conv_wgrad(1, g, dy, x0, x1, gw, ws=ws); torch.cuda.synchronize()
    a = stamps(1); n = len(a); m = a.mean(0)
    print(f"{c0}+{c1}->{co}@{H} k{k}: waves {n} slabs/wave {m[4]:.1f} | entry->loop {m[8]:.0f} | loop {m[6]:.0f} ({m[6]/max(m[4],1):.0f}/slab) | loop end->stores done {m[9]:.0f} cyc")

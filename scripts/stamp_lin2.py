#!/usr/bin/env python3
"""Phase cycles of conv_lin2's slab loop (library built with `make EXTRA=-DMDM_STAMP`)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import torch
from mdm import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
fn = lib.mdm_debug_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
import numpy as np
def stamps(reset=1):
    buf = (ctypes.c_ulonglong * (4096 * 32))(); assert fn(buf, reset) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 32).astype(np.float64)
    return a[a[:, 5] > 0]
shapes = [(256, 0, 256, 8, 1), (256, 256, 256, 8, 1), (256, 0, 768, 8, 1), (256, 0, 256, 16, 1), (128, 0, 128, 32, 3)]
for (c0, c1, co, H, KS) in shapes:
    pd = KS // 2
    g = ops.ConvGeom(N=32, IH=H, IW=H, C0=c0, C1=c1, Cout=co, KH=KS, KW=KS, stride=1, pad_t=pd, pad_l=pd, pad_b=pd, pad_r=pd, ups=0)
    bf = torch.bfloat16
    x0 = torch.randn(32, H, H, c0, device=dev, dtype=bf)
    w = torch.randn(KS * KS, co, c0 + c1, device=dev, dtype=bf) * 0.02
    x1 = torch.randn(32, H, H, c1, device=dev, dtype=bf) if c1 else None
    b = torch.zeros(co, device=dev); y = torch.empty(32, H, H, co, device=dev, dtype=bf)
    ws = torch.empty(16 * 9 * 256 * 512, device=dev)
    for _ in range(3): ops.conv_fwd(1, g, x0, x1, w, b, y, ws=ws)
    torch.cuda.synchronize(); stamps(1)
    reps = 1
    for _ in range(reps): ops.conv_fwd(1, g, x0, x1, w, b, y, ws=ws)
    torch.cuda.synchronize()
    a = stamps(1); st = a.sum(0); tw, tb, ti, tc, nk, nw, tot, _ = st[:8]
    span = (a[:, 7] + a[:, 6]).max() - a[:, 7].min()
    print(f"{c0}+{c1}->{co}@{H} k{KS}: waves/launch {nw/reps:.0f} slabs/wave {nk/nw:.1f} | per slab per wave: [non-pipe: vmcnt-wait|barrier|issue|reads+mfma; pipe: phaseA|barrier|waits|phaseB] {tw/nk:.0f} {tb/nk:.0f} {ti/nk:.0f} {tc/nk:.0f} cyc | loop total/wave {tot/nw:.0f} cyc; first loop start -> last loop end {span:.0f} cyc")
    print(f"   entry->loop {st[8]/nw:.0f} cyc, loop {st[6]/nw:.0f}, loop end->stores done {st[9]/nw:.0f}; first entry -> last done {(a[:,10]+a[:,8]+a[:,6]+a[:,9]).max()-a[:,10].min():.0f} cyc; entry spread {a[:,10].max()-a[:,10].min():.0f}")
    continue
    print("   issue per iteration:", [round(v / nw) for v in st[8:20]], " barrier per iteration:", [round(v / nw) for v in st[20:32]])

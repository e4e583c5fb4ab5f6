#!/usr/bin/env python3
"""Per-role cycles of conv_halo_ws_body (library built with `make EXTRA=-DMDM_STAMP`, selected with MDM_LIB_PATH): the loading waves
(counted-vmcnt wait, barrier, halo split) and the multiplying waves (barrier) separately, per tap.
    python scripts/stamp_halo_ws.py [N=100]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import numpy as np, torch
from mdm import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
fn = lib.mdm_debug_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
NREC = 4096
def stamps(reset=1):
    buf = (ctypes.c_ulonglong * (NREC * 32))(); assert fn(buf, reset) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(NREC, 32).astype(np.float64)
    return a
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for (c0, c1, co, H) in [(128, 0, 128, 32), (256, 0, 256, 16), (256, 256, 256, 16)]:
    g = ops.ConvGeom(N=N, IH=H, IW=H, C0=c0, C1=c1, Cout=co)
    x0 = torch.randn(N, H, H, c0, device=dev); x1 = torch.randn(N, H, H, c1, device=dev) if c1 else None
    w = torch.randn(9, co, c0 + c1, device=dev) * 0.02
    ws = torch.empty_like(w); segs = torch.tensor([[0, w.numel()]], dtype=torch.int64, device=dev)
    _lib.call("mdm_split_shadow", _lib.ptr(w), _lib.ptr(ws), _lib.ptr(segs), 1, _lib.stream())
    b = torch.zeros(co, device=dev); y = torch.empty(N, H, H, co, device=dev)
    for _ in range(2): ops.conv_fwd(0, g, x0, x1, w, b, y, w_split=ws)
    torch.cuda.synchronize(); stamps(1)
    ops.conv_fwd(0, g, x0, x1, w, b, y, w_split=ws); torch.cuda.synchronize()
    a = stamps(1)
    idx = np.arange(NREC)
    live = a[:, 5] > 0
    for tag, sel in (("multiplying", live & ((idx % 8) < 4)), ("loading", live & ((idx % 8) >= 4))):
        r = a[sel]
        if not len(r): continue
        ntap = r[:, 4].mean()
        print(f"{c0}+{c1}->{co}@{H} N={N} {tag:12s} waves {len(r):4d}: per tap: loop {r[:,6].mean()/ntap:6.0f} cyc (vmcnt wait {r[:,0].mean()/ntap:5.0f}, barrier {r[:,1].mean()/ntap:5.0f}, "
              f"halo split {r[:,2].mean()/ntap:5.0f}); entry->loop {r[:,8].mean():6.0f}, tail {r[:,9].mean():6.0f}")

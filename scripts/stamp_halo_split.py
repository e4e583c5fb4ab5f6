#!/usr/bin/env python3
"""Phase cycles of the fp32 halo kernels (exact / split products) on the sampler's layers (library built with `make EXTRA=-DMDM_STAMP`,
selected with MDM_LIB_PATH)."""
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
    return a[a[:, 5] > 0]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for (c0, c1, co, H) in [(128, 0, 128, 32), (256, 0, 256, 16), (256, 256, 256, 16), (256, 0, 256, 8), (512, 0, 512, 4)]:
    g = ops.ConvGeom(N=N, IH=H, IW=H, C0=c0, C1=c1, Cout=co)
    x0 = torch.randn(N, H, H, c0, device=dev); x1 = torch.randn(N, H, H, c1, device=dev) if c1 else None
    w = torch.randn(9, co, c0 + c1, device=dev) * 0.02
    ws = torch.empty_like(w); segs = torch.tensor([[0, w.numel()]], dtype=torch.int64, device=dev)
    _lib.call("mdm_split_shadow", _lib.ptr(w), _lib.ptr(ws), _lib.ptr(segs), 1, _lib.stream())
    b = torch.zeros(co, device=dev); y = torch.empty(N, H, H, co, device=dev)
    for tag, sp in (("exact", None), ("split", ws)):
        for _ in range(2): ops.conv_fwd(0, g, x0, x1, w, b, y, w_split=sp)
        torch.cuda.synchronize(); stamps(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv_fwd(0, g, x0, x1, w, b, y, w_split=sp); e1.record(); torch.cuda.synchronize()
        a = stamps(1); st = a.sum(0); nw = st[5]; ntap = st[4]
        print(f"{c0}+{c1}->{co}@{H} N={N} {tag}: {e0.elapsed_time(e1)*1e3:.1f} us, waves recorded {nw:.0f}; per tap per wave: loop {st[6]/ntap:.0f} cyc "
              f"(vmcnt wait {st[0]/ntap:.0f}, barrier {st[1]/ntap:.0f}, halo split {st[2]/ntap:.0f}); entry->loop {st[8]/nw:.0f}, loop {st[6]/nw:.0f}, tail {st[9]/nw:.0f}")

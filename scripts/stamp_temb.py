#!/usr/bin/env python3
"""The time-embedding path's skinny fp32 layers (mdm_skinny_linear_fwd) at the bench shapes: graph-timed per launch and, with a stamped
build (`make EXTRA=-DMDM_STAMP`, MDM_LIB_PATH), cycles of load / multiply / meet per workgroup.  python scripts/stamp_temb.py [M=32]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import numpy as np, torch
from mdm import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
fn = getattr(lib, "mdm_debug_stamps_temb", None)
if fn is not None: fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
def stamps():
    buf = (ctypes.c_ulonglong * (1024 * 8))(); assert fn(buf, 1) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
    return a[a[:, 0] > 0]
def ev():
    e = ctypes.c_void_p(); _lib.check(lib.mdm_event_create(ctypes.byref(e))); return e
def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    with _lib.Recording() as rec:
        for _ in range(reps): f()
    gx = _lib.GraphExec(rec); gx.launch(); torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream; a, b = ev(), ev()
    lib.mdm_event_record(a, st); gx.launch(); lib.mdm_event_record(b, st)
    ms = ctypes.c_float(); _lib.check(lib.mdm_event_elapsed_ms(a, b, ctypes.byref(ms)))
    return ms.value * 1e3 / reps
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for (K, N, temb, act) in [(128, 512, True, True), (512, 512, False, True), (512, 4992, False, False)]:
    W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    x = None if temb else torch.randn(M, K, device=dev)
    t = torch.randint(1, 1000, (M,), device=dev).float() if temb else None
    y = torch.empty(M, N, device=dev); a = torch.empty(M, N, device=dev) if act else None
    f = lambda: ops.skinny_linear_fwd(x, W, b, M, N, K, y, act_out=a, t=t)
    us = timeit(f)
    line = f"M={M} K={K} N={N} temb={int(temb)}: {us:6.1f} us per launch ({N * K * 4 / 1e6:.1f} MB of weights)"
    if fn is not None:
        f(); torch.cuda.synchronize(); stamps(); f(); torch.cuda.synchronize(); s = stamps()
        if len(s):
            m = s.mean(0)
            line += f" | per workgroup: entry->loads landed {m[1]:.0f}, multiply {m[2]:.0f}, meet {m[3]:.0f} cyc; span {s[:,5].max() - s[:,4].min():.0f} cyc over {len(s)} workgroups"
    print(line, flush=True)

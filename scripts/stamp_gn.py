#!/usr/bin/env python3
"""Phase cycles of gn_bwd_reg_kernel (library built with `make EXTRA=-DMDM_STAMP`) + graph-timed GN fwd/bwd per shape."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import torch, numpy as np
from mdm import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
fn = getattr(lib, "mdm_debug_stamps_norm", None)
if fn is not None: fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
def stamps(reset=1):
    buf = (ctypes.c_ulonglong * (4096 * 16))(); assert fn(buf, reset) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).astype(np.float64)
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
N = 32
for (C, H) in [(128, 32), (256, 16), (256, 8), (256, 4), (384, 32), (512, 16)]:
    P = H * H
    bf = torch.bfloat16
    x = torch.randn(N, P, C, device=dev, dtype=bf); dy = torch.randn_like(x); y = torch.empty_like(x); dx = torch.empty_like(x)
    ga = torch.randn(C, device=dev); be = torch.randn(C, device=dev); st = torch.empty(N, 32, 2, device=dev)
    dga = torch.zeros(C, device=dev); dbe = torch.zeros(C, device=dev); ws = torch.empty(N * 64 * 4, device=dev)
    f = lambda: ops.groupnorm_fwd(1, x, C, None, 0, N, P, ga, be, True, y, st, ws)
    b = lambda: ops.groupnorm_bwd(1, x, C, None, 0, N, P, ga, be, True, dy, st, dx, 0, None, 0, dga, dbe, ws)
    tf, tb = timeit(f), timeit(b)
    mb = N * P * C * 2 / 1e6
    print(f"C={C} {H}x{H}: tensor {mb:.2f} MB | fwd {tf:.1f} us ({2*mb/tf*1e-3:.2f} TB/s) | bwd {tb:.1f} us ({3*mb/tb*1e-3:.2f} TB/s)")
    if fn is not None:
        b(); torch.cuda.synchronize(); stamps(1); b(); torch.cuda.synchronize(); a = stamps(1)
        if len(a):
            m = a.mean(0)
            print(f"    bwd per wave: setup+loads {m[1]:.0f} | compute+shuffle+lds-atomics {m[2]:.0f} | sync {m[3]:.0f} | global atomics {m[4]:.0f} | apply+store {m[5]:.0f} cyc; kernel span {a[:,7].max()-a[:,6].min():.0f} cyc, waves {len(a)}")

#!/usr/bin/env python3
"""Where the time of the grouped weight-gradient launches goes (library built with `make EXTRA=-DMDM_STAMP`): one record per workgroup
(= work item) with prologue / loop / epilogue cycles, absolute start / end and the CU it ran on.  Per launch: the span, the busy share of
the CUs, the split of the busy time, cycles per slab by tile kind.   python scripts/stamp_group.py [--batch 32]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import numpy as np, torch
import mdm
from mdm import _lib
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); o = ap.parse_args()
lib = _lib.load()
fn = lib.mdm_debug_stamps_n
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
NREC = 32768
def stamps():
    buf = (ctypes.c_ulonglong * (NREC * 32))(); assert fn(buf, NREC, 1) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(NREC, 32).astype(np.float64)
    return a[a[:, 5] > 0]
model = mdm.UNet(mdm.unet6_config(32), N=o.batch, H=32, W=32, dtype=mdm.BF16, seed=0, use_graph=False)
st = torch.cuda.current_stream().cuda_stream
model.zero_grad(); model.forward_plan.run(st); model.backward_plan.run(st); torch.cuda.synchronize()
for i, (name, f, args) in enumerate(model.backward_plan.calls):
    if name != "mdm_wgrad_group_launch":
        continue
    for _ in range(2):
        f(*args, st)
    torch.cuda.synchronize(); stamps()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); f(*args, st); ev1.record(); torch.cuda.synchronize()
    a = stamps()
    t0, t1 = a[:, 10].min(), a[:, 16].max()
    span = t1 - t0
    busy = (a[:, 16] - a[:, 10])
    hw = a[:, 15].astype(np.uint64)
    cu = ((hw >> np.uint64(8)) & np.uint64(0xFF)) | ((hw >> np.uint64(32)) << np.uint64(8))      # (cu, sh, se) of HW_ID | XCC_ID
    ncu = len(np.unique(cu))
    print(f"bwd call {i}: {len(a)} items on {ncu} CUs, event {ev0.elapsed_time(ev1) * 1e3:.1f} us, span {span:.0f} ticks; "
          f"sum(item time) / (CUs x span) = {busy.sum() / (ncu * span):.2f}")
    print(f"    of the item time: prologue {a[:, 8].sum() / busy.sum():.2f}  loop {a[:, 6].sum() / busy.sum():.2f}  epilogue {a[:, 9].sum() / busy.sum():.2f}")
    for bm in sorted(set(a[:, 11])):
        s = a[a[:, 11] == bm]
        print(f"    tile BM={int(bm):3d}: {len(s):5d} items, slabs/item {s[:, 4].mean():6.1f}, loop ticks/slab {s[:, 6].sum() / s[:, 4].sum():7.0f}, "
              f"prologue {s[:, 8].mean():6.0f}, epilogue {s[:, 9].mean():6.0f}, share of item time {(s[:, 16] - s[:, 10]).sum() / busy.sum():.2f}")
    f9 = a[a[:, 11] == 999]
    if len(f9):
        ns = f9[:, 4].sum()
        print(f"    nine-tap loop per slab: wait+barrier {f9[:, 0].sum() / ns:.0f}  transpose {f9[:, 1].sum() / ns:.0f}  compute {f9[:, 2].sum() / ns:.0f} ticks")
        print(f"                   wave 4: wait+barrier {f9[:, 19].sum() / ns:.0f}  transpose {f9[:, 20].sum() / ns:.0f}  compute {f9[:, 21].sum() / ns:.0f} ticks")
    # per-CU: when does each CU finish relative to the span (tail imbalance)
    ends = np.array([a[cu == c, 16].max() for c in np.unique(cu)]) - t0
    per_cu_busy = np.array([busy[cu == c].sum() for c in np.unique(cu)])
    print(f"    CU finish times / span: min {ends.min() / span:.2f} median {np.median(ends) / span:.2f};  per-CU busy / span: min {per_cu_busy.min() / span:.2f} "
          f"median {np.median(per_cu_busy) / span:.2f} max {per_cu_busy.max() / span:.2f}")
    # by layer shape
    keys = sorted(set(zip(a[:, 12], a[:, 13], a[:, 14], a[:, 17], a[:, 18])))
    for k in keys:
        s = a[(a[:, 12] == k[0]) & (a[:, 13] == k[1]) & (a[:, 14] == k[2]) & (a[:, 17] == k[3]) & (a[:, 18] == k[4])]
        print(f"      M{int(k[0]):4d} N{int(k[1]):4d} K{int(k[2]):6d} taps{int(k[3])} sk{int(k[4]):2d}: {len(s):4d} items, ticks/slab {s[:, 6].sum() / max(s[:, 4].sum(), 1):6.0f}, "
              f"pro {s[:, 8].mean():6.0f} epi {s[:, 9].mean():6.0f}, item ticks {(s[:, 16] - s[:, 10]).mean():8.0f}")

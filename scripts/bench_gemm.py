#!/usr/bin/env python3
"""Per-shape timing of the contraction kernel on the conv sites of unet6 @32x32, N=32 (HIP events)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import torch
from mdm import _lib, ops
dev = torch.device("cuda:0")
N = int(os.environ.get("BN", "32"))
shapes = [  # (Cin0, Cin1, Cout, H, k, stride, ups, count per step)
    (128, 0, 128, 32, 3, 1, 0, 7), (128, 128, 128, 32, 3, 1, 0, 2), (256, 128, 128, 32, 3, 1, 0, 1), (256, 0, 256, 16, 3, 1, 1, 1),
    (128, 0, 256, 16, 3, 1, 0, 1), (256, 0, 256, 16, 3, 1, 0, 7), (256, 256, 256, 16, 3, 1, 0, 2), (256, 128, 256, 16, 3, 1, 0, 1),
    (256, 0, 256, 8, 3, 1, 0, 8), (256, 256, 256, 8, 3, 1, 0, 3), (256, 0, 256, 4, 3, 1, 0, 11), (256, 256, 256, 4, 3, 1, 0, 3),
    (128, 0, 128, 32, 3, 2, 0, 1), (256, 0, 256, 16, 3, 2, 0, 1), (256, 0, 256, 8, 3, 2, 0, 1),
    (256, 0, 768, 8, 1, 1, 0, 5), (256, 0, 256, 8, 1, 1, 0, 5), (256, 256, 256, 8, 1, 1, 0, 3), (128, 128, 128, 32, 1, 1, 0, 2),
]
def ev():
    e = ctypes.c_void_p(); _lib.check(_lib.load().mdm_event_create(ctypes.byref(e))); return e
def timeit(fn, reps=20):
    """GPU time per launch: the launches are recorded, captured into a hipGraph and the replay is timed
    (eager ctypes launches are host-bound for the small layers)."""
    fn(); torch.cuda.synchronize()
    with _lib.Recording() as rec:
        for _ in range(reps): fn()
    gx = _lib.GraphExec(rec)
    gx.launch(); torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    a, b = ev(), ev()
    _lib.load().mdm_event_record(a, st)
    gx.launch()
    _lib.load().mdm_event_record(b, st)
    ms = ctypes.c_float(); _lib.check(_lib.load().mdm_event_elapsed_ms(a, b, ctypes.byref(ms)))
    return ms.value * 1e3 / reps
WS = torch.empty(16 * 9 * 256 * 512, device=dev) if os.environ.get("NOWS") != "1" else None
sel = os.environ.get("SHAPES")
if sel:
    shapes = [shapes[int(i)] for i in sel.split(",")]
PASSES = os.environ.get("PASSES", "fwd,dgrad,wgrad").split(",")
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}; totf = 0.0
print(f"{'shape':38s} {'GF':>6s} | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}")
for (c0, c1, co, H, k, s, ups, cnt) in shapes:
    pads = (1, 1, 1, 1) if k == 3 and s == 1 else (0, 0, 1, 1) if k == 3 else (0, 0, 0, 0)
    g = ops.ConvGeom(N=N, IH=H >> ups if ups else H, IW=H >> ups if ups else H, C0=c0, C1=c1, Cout=co, KH=k, KW=k, stride=s,
                     pad_t=pads[0], pad_l=pads[1], pad_b=pads[2], pad_r=pads[3], ups=ups)
    bf = torch.bfloat16
    x0 = torch.randn(N, g.IH, g.IW, c0, device=dev, dtype=bf)
    x1 = torch.randn(N, g.IH, g.IW, c1, device=dev, dtype=bf) if c1 else None
    w = torch.randn(k * k, co, c0 + c1, device=dev, dtype=bf) * 0.02
    b = torch.zeros(co, device=dev)
    y = torch.empty(N, g.OH, g.OW, co, device=dev, dtype=bf)
    dy = torch.randn_like(y)
    gx0 = torch.empty(N, g.VH, g.VW, c0, device=dev, dtype=bf)
    gx1 = torch.empty(N, g.VH, g.VW, c1, device=dev, dtype=bf) if c1 else None
    gw = torch.zeros(k * k, co, c0 + c1, device=dev)
    fl = ops.conv_flops(g)
    t_f = timeit(lambda: ops.conv_fwd(1, g, x0, x1, w, b, y, ws=WS)) if "fwd" in PASSES else 1e9
    wT = w.transpose(1, 2).contiguous()
    if ups:        # folded upsample: the general gather path (as in the model)
        t_d = timeit(lambda: ops.conv_dgrad_t(1, g, dy, wT, torch.empty(N, g.VH, g.VW, c0, device=dev, dtype=bf), 0, ws=WS)) if "dgrad" in PASSES else 1e9
    else:
        t_d = timeit(lambda: ops.conv_dgrad_t(1, g, dy, wT, gx0, 0, gx1, 0, ws=WS)) if "dgrad" in PASSES else 1e9
    t_w = timeit(lambda: ops.conv_wgrad(1, g, dy, x0, x1, gw, ws=WS)) if "wgrad" in PASSES else 1e9
    name = f"{c0}+{c1}->{co} @{H} k{k} s{s}{' up' if ups else ''} x{cnt}"
    print(f"{name:38s} {fl/1e9:6.2f} | {t_f:8.1f} {fl/t_f/1e6:6.1f} | {t_d:8.1f} {fl/t_d/1e6:6.1f} | {t_w:8.1f} {fl/t_w/1e6:6.1f}")
    tot["fwd"] += t_f * cnt; tot["dgrad"] += t_d * cnt; tot["wgrad"] += t_w * cnt; totf += fl * cnt
print("weighted per-step totals (us):", {k: round(v) for k, v in tot.items()}, "sum", round(sum(tot.values())), f"-> {3*totf/sum(tot.values())/1e6:.1f} TF/s avg")

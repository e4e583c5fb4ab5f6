import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from golden.make_golden import TINY, base_args, seed_all
import mdm
from oracle.unet_ref import random_params
g = np.load(os.path.join(ROOT, "tests/golden/sampler.npz"))
use_graph = (sys.argv[2] == "1") if len(sys.argv) > 2 else True
order = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(int(g["samp_n"])))
model = mdm.UNet(TINY, N=2, H=16, W=16, dtype=int(sys.argv[1]) if len(sys.argv) > 1 else 0, params=random_params(TINY), use_graph=use_graph).eval()
P0 = model.store.P.clone()
names = ["sample_t", "shift", "shifted", "mask", "shifted_result", "sample_0", "degraded_mask", "degraded_mask_next", "degraded_t", "difference", "degraded_next_t"]
for i in order:
    dep, mode, sel, ch, kind, st = [None if str(x) == "None" else str(x) for x in g[f"samp{i}_cfg"]]
    a = base_args(data_size=16, ddpm_schedule=kind, ddpm_num_steps=6, select_degrade_pixel=sel, degrade_channel=ch, shift_type=st,
                  sampling_mask_dependency=dep, momentum_adaptive=mode, sample_num=2, sample_latent_shape="uniform")
    s = mdm.Scheduler(a); s.update_ddpm_num_steps(6); ts = s.get_timesteps_epoch(0, 1)
    smp = mdm.Sampler(None, a, s, [None] * 3)
    seed_all(400 + i)
    x0, hist = smp.sample(model, ts)
    ref = g[f"samp{i}_hist"]
    print("cfg", i, dep, mode, sel, kind, st, "T", len(ts), "weights changed:", float((model.store.P - P0).abs().max()))
    for slot in range(1, len(ts) + 1):
        row = []
        for j in range(11):
            d = np.linalg.norm(hist[j][slot].numpy() - ref[j][slot]); n = np.linalg.norm(ref[j][slot])
            row.append(f"{names[j][:6]}:{d/(n+1e-30):.1e}/{n:.1e}")
        print("  slot", slot, " ".join(row[3:4]))

import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "masked-diffusion-model_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, mdm
from golden.make_golden import TINY, base_args, seed_all
from oracle.unet_ref import random_params
g = np.load(os.path.join(ROOT, "tests/golden/sampler_dep_t.npz"))
model = mdm.UNet(TINY, N=2, H=16, W=16, dtype=0, params=random_params(TINY)).eval()
for i in range(int(g["dept_n"])):
    mode, ch, kind, st, mo, ma = [str(v) for v in g[f"dept{i}_cfg"]]
    a = base_args(data_size=16, ddpm_schedule=kind, ddpm_num_steps=8, select_degrade_pixel="thresholding", degrade_channel=ch,
                  shift_type=st, sampling_mask_dependency="dependent_t", momentum_adaptive=mode, sample_num=2,
                  sample_latent_shape="uniform", mean_option=mo, mean_area=ma, noise_mean=0.05)
    s = mdm.Scheduler(a); s.update_ddpm_num_steps(8); ts = s.get_timesteps_epoch(0, 1)
    seed_all(800 + i)
    x0, hist = mdm.Sampler(None, a, s, [None] * 3).sample(model, ts)
    ref = g[f"dept{i}_hist"]
    for j in (0, 1, 6, 7):
        h = hist[j].numpy()
        for slot in range(h.shape[0]):
            eq = np.array_equal(h[slot], ref[j][slot], equal_nan=True)
            if not eq:
                d = np.abs(h[slot] - ref[j][slot])
                print(i, mode, st, "hist", j, "slot", slot, "max diff", np.nanmax(d), "n diff", int((d > 0).sum()), "nan", int(np.isnan(ref[j][slot]).sum()), int(np.isnan(h[slot]).sum()))

import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/masked-diffusion-model_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from mdm import unet as U
import mdm
from oracle.unet_ref import random_params
from golden.make_golden import TINY, base_args
g = np.load("/root/repo/tests/golden/unet.npz")
cfg = U.unet6_config(32)
def rel(a, b): return float((a.float().cpu() - torch.as_tensor(b).float()).norm() / torch.as_tensor(b).float().norm())
for n in (1, 5):
    exact = U.UNet(cfg, N=n, H=32, W=32, dtype=0, params=random_params(cfg, 77)).eval()
    split = U.UNet(cfg, N=n, H=32, W=32, dtype=0, store=exact.store, f32_products="split").eval()
    x = torch.from_numpy(g["unet32_x"]).repeat(n, 1, 1, 1) + 0.25 * torch.arange(n).view(n, 1, 1, 1)
    t = torch.from_numpy(g["unet32_t"]).repeat(n) + 7 * torch.arange(n)
    ye = exact(x, t).sample.clone(); ys = split(x, t).sample.clone(); torch.cuda.synchronize()
    print("preset n", n, "split vs exact", rel(ys, ye.cpu()), "exact vs golden", rel(ye[:1], g["unet32_y"]), "split vs golden", rel(ys[:1], g["unet32_y"]))

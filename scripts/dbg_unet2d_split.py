import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/masked-diffusion-model_amd")
import torch
from mdm.unet2d import UNet2D, my_model_config
cfg = my_model_config(3, 32, 1)
m = UNet2D(cfg, N=2, H=32, W=32, dtype=1, seed=3)
x = torch.randn(4, 3, 32, 32); t = torch.tensor([5.0, 100.0, 500.0, 999.0])
pe = m.sampling_plan(4, "f32").eval(); ps = m.sampling_plan(4, "f32_split").eval()
ye = pe(x, t).sample.clone(); ys = ps(x, t).sample.clone(); torch.cuda.synchronize()
print("unet2d split vs exact", float((ys - ye).norm() / ye.norm()), "finite", bool(torch.isfinite(ys).all()))

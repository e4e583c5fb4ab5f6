import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import torch
from mdm import ops
from mdm.unet2d import UNet2D, default_init_params, my_model_config
from oracle.unet2d_ref import unet2d_forward
hw, natt, dt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
boc = tuple(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else (32, 32, 64, 64, 96)
cfg = my_model_config(3, hw, natt, block_out_channels=boc)
n = 2
p = default_init_params(UNet2D(cfg, n, hw, hw, _dry=True).reference_shapes(), 5)
g = torch.Generator().manual_seed(17)
x = torch.rand(n, 3, hw, hw, generator=g) * 2 - 1
t = torch.tensor([4.0, 777.0]); gy = torch.randn(n, 3, hw, hw, generator=g)
net = UNet2D(cfg, N=n, H=hw, W=hw, dtype=dt, params=p)
y = net(x, t).sample
net.zero_grad()
ops.nchw_to_nhwc(dt, gy.to(net.device), net.y_out.grad, n, 3, hw, hw, net.cout_p)
net.run_backward(); torch.cuda.synchronize()
q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
yo = unet2d_forward(q, cfg, x, t); (yo * gy).sum().backward()
rel = lambda a, b: float((a.float().cpu() - b).norm() / (b.norm() + 1e-20))
print("y", rel(y, yo.detach()))
grads = net.store.grad_dict()
for k in net.reference_param_order():
    r = rel(grads[k], q[k].grad)
    if r > 0.15: print(f"{r:8.3f} {float(q[k].grad.norm()):10.3e} {k}")

import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from golden.make_golden import TINY, base_args, seed_all
import mdm
from mdm import ops
from oracle.unet_ref import random_params, UNetRef
g = np.load(os.path.join(ROOT, "tests/golden/train_step.npz"))
a = base_args(data_size=16, ddpm_schedule="log", ddpm_num_steps=10, select_degrade_pixel="indexing", degrade_channel=None, shift_type="non_shift", batch_size=4)
model = mdm.UNet(TINY, N=4, H=16, W=16, dtype=0, params=random_params(TINY))
opt = mdm.AdamW(model, lr=1e-3)
tr = mdm.BaseTrainer(a, None, None, model, None, opt, mdm.get_lr_scheduler("constant", opt, 0, 10), mdm.Accelerator())
tr.Scheduler.update_ddpm_num_steps(10); tr.timesteps_used_epoch = tr.Scheduler.get_timesteps_epoch(0, 1)
seed_all(500)
r = tr._run_batch(0, (torch.from_numpy(g["step_x0"]), None, None), 0, 1, 0, None, None)
pred = torch.empty(4, 3, 16, 16, device="cuda")
ops.nhwc_to_nchw(0, model.y_out.data, pred, 4, 3, 16, 16, model.cout_p)
ref = g["step_base_pred"]
t = tr.step.last["t"]
print("t", t, "loss", r[0], float(g["step_base_loss"]))
for n in range(4):
    e = np.linalg.norm(pred[n].cpu().numpy() - ref[n]) / np.linalg.norm(ref[n])
    print(n, "rel", e, "x_in zero frac", float((tr.step.x_in[n] == 0).float().mean()))
o = UNetRef(TINY)
with torch.no_grad():
    yo = o(tr.step.x_in.cpu(), t.float()).sample
print("oracle vs golden", float((yo - torch.from_numpy(ref)).norm() / torch.from_numpy(ref).norm()))
# layer-by-layer: compare first activations
from oracle import unet_ref as R
p = o.pdict()
with torch.no_grad():
    h0 = R._conv(tr.step.x_in.cpu(), p, "in_conv", padding=1)
    a1 = torch.nn.functional.silu(R._gn(h0, p, "downsamples.level_0.0.norm1"))
acts = {a.name: a for a in model.acts}
mine_h0 = acts["in_conv"].data.permute(0, 3, 1, 2).cpu()
mine_a1 = acts["downsamples.level_0.0.norm1"].data.permute(0, 3, 1, 2).cpu()
for n in range(4):
    print(n, "in_conv", float((mine_h0[n] - h0[n]).abs().max()), "norm1", float((mine_a1[n] - a1[n]).abs().max()), float(a1[n].abs().max()))

"""Per-step (teacher-forced) error of the fp32 HIP sampler against the oracle along a T-step schedule: where on the
trajectory the kernels are least accurate.  Usage: python scripts/tf_error_profile.py [T] [n]  (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "masked-diffusion-model_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import mdm
from golden.make_golden import TINY, base_args, seed_all
from oracle.sampler_ref import SamplerRef
from oracle.scheduler_ref import SchedulerRef
from oracle.unet_ref import UNetRef, random_params

T = int(sys.argv[1]) if len(sys.argv) > 1 else 250
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
hw = 16
a = base_args(data_size=hw, ddpm_schedule="linear", ddpm_num_steps=T, shift_type="noise_with_perturbation",
              sampling_mask_dependency="independent", momentum_adaptive="base_momentum", sample_num=n,
              sample_latent_shape="uniform", sample_history="device")
params = random_params(TINY)
rs = SchedulerRef(a); rs.update_ddpm_num_steps(T); ts = rs.get_timesteps_epoch(0, 1)
seed_all(4250)
want, ref = SamplerRef(None, a, rs, [None] * 3).sample(UNetRef(TINY, params), ts)
ref = dict(zip(mdm.sampler.HISTORY_NAMES, ref))
m64 = UNetRef(TINY, params, dtype=torch.float64)
# exact (fp64) network output at the oracle's own inputs of every step
with torch.no_grad():
    exact = torch.stack([m64(ref["shifted"][k], torch.full((n,), float(ts[T - k]))).sample.float() for k in range(1, T + 1)])
ref_xt = ref["sample_t"].cuda()
model = mdm.UNet(TINY, N=n, H=hw, W=hw, dtype=0, params=params).eval()
s = mdm.Scheduler(a); s.update_ddpm_num_steps(T)
smp = mdm.Sampler(None, a, s, [None] * 3)
smp.step_hook = lambda i, slot, x_t: x_t.copy_(ref_xt[slot])
seed_all(4250)
x0, hist = smp.sample(model, ts)
hist = {k: v.cpu() for k, v in zip(mdm.sampler.HISTORY_NAMES, hist)}
pred_h, pred_r = hist["mask"][1:].double().flatten(1), ref["mask"][1:].double().flatten(1)
ex = exact.double().flatten(1)
e_hip = (pred_h - ex).norm(dim=1) / ex.norm(dim=1)
e_cpu = (pred_r - ex).norm(dim=1) / ex.norm(dim=1)
print("step  |x_t|  err(hip vs fp64)  err(cpu fp32 vs fp64)")
for k in list(range(0, 12)) + list(range(12, T, max(1, T // 25))):
    print(f"{k + 1:5d} {float(ref['sample_t'][k + 1].norm()):10.2f} {float(e_hip[k]):.3e} {float(e_cpu[k]):.3e}")
print("median hip", float(e_hip.median()), "cpu", float(e_cpu.median()), "max hip", float(e_hip.max()), "at", int(e_hip.argmax()) + 1,
      "max cpu", float(e_cpu.max()), "at", int(e_cpu.argmax()) + 1)
print("mean ratio hip/cpu", float((e_hip / e_cpu).mean()))

"""Wall-clock of the reverse sampler per step for one storage type / product mode (sample_num images, `steps` reverse steps):
    python scripts/sampler_time.py [f32|f32_split|bf16] [steps] [sample_num]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import torch, mdm
from bench import make_args
mode = sys.argv[1] if len(sys.argv) > 1 else "f32_split"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
a = make_args()
a.sample_num = n
model = mdm.UNet(mdm.unet6_config(32), N=n, H=32, W=32, dtype=mdm.BF16 if mode == "bf16" else mdm.F32, seed=0,
                 f32_products="split" if mode == "f32_split" else "exact").eval()
S = mdm.Scheduler(a); S.update_ddpm_num_steps(1000); used = S.get_timesteps_epoch(0, 1)
smp = mdm.Sampler(None, a, S, [None] * 3)
smp.sample(model, used[:3]); torch.cuda.synchronize()
t0 = time.perf_counter(); x0, _ = smp.sample(model, used[:steps]); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"sampler {mode} N={n}: {1e3 * (t1 - t0) / steps:.3f} ms/step  finite={bool(torch.isfinite(x0).all())}")

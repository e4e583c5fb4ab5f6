import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import torch, mdm
from bench import make_args
a = make_args()
model = mdm.UNet(mdm.unet6_config(32), N=100, H=32, W=32, dtype=mdm.BF16, seed=0)
S = mdm.Scheduler(a); S.update_ddpm_num_steps(1000); used = S.get_timesteps_epoch(0, 1)
smp = mdm.Sampler(None, a, S, [None] * 3)
smp.sample(model, used[:3]); torch.cuda.synchronize()
t0 = time.perf_counter(); smp.sample(model, used[:200]); torch.cuda.synchronize(); t1 = time.perf_counter()
print("sampler ms/step", 1e3 * (t1 - t0) / 200)
t0 = time.perf_counter()
for _ in range(200): model.run_forward()
torch.cuda.synchronize(); t1 = time.perf_counter()
print("forward-only graph ms/step", 1e3 * (t1 - t0) / 200)
# host-only cost: run the loop without waiting (queue depth) is what matters; measure python time per step w/o sync
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); smp.sample(model, used[:100]); pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(12)

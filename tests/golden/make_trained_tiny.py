"""Trains the TINY unet6 (tests/golden/make_golden.py) with the CPU ORACLE's train step on structured synthetic images and
stores the weights as a fixture (tests/golden/trained_tiny.npz): the net the free-running 1000-step sampler parity test runs on.

Why: an UNTRAINED U-Net iterated 1000 times is a chaotic map -- the oracle's own fp32 run ends 0.75 (rel-L2) away from its own
fp64 run, so north_star's "sampler output within 1e-3 of the CPU reference" could only be checked step by step
(teacher forcing).  A trained denoiser is contractive; on it the end-to-end claim can be asserted.

    python tests/golden/make_trained_tiny.py [steps]        (CPU, ~3 min for 600 steps; deterministic for a fixed thread count)

The images are smooth colour fields (a few low-frequency cosines per channel with random phase + a random mean colour), in [-1, 1]
like the reference's `Normalize([0.5], [0.5])` data (utils/mydataset.py:81).  Training follows reference
trainer_masked_mean_shift.py:82-193 through oracle/trainer_ref.py (linear T = 1000 schedule, thresholding masks,
noise_with_perturbation shifts, clip 1.0, AdamW).  The file also records the yardstick of the finished net: the oracle's fp32
free run against the same loop with an fp64 network.
"""
import math
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from golden.make_golden import TINY, base_args, seed_all  # noqa: E402

HW, T = 16, 1000


def smooth_images(n, gen):
    """[n, 3, HW, HW] in [-1, 1]: mean colour + three low-frequency cosine waves per channel."""
    yy, xx = torch.meshgrid(torch.arange(HW, dtype=torch.float32), torch.arange(HW, dtype=torch.float32), indexing="ij")
    img = torch.rand(n, 3, 1, 1, generator=gen) * 1.2 - 0.6
    for _ in range(3):
        fx = torch.randint(0, 3, (n, 3, 1, 1), generator=gen).float()
        fy = torch.randint(0, 3, (n, 3, 1, 1), generator=gen).float()
        ph = torch.rand(n, 3, 1, 1, generator=gen) * 2 * math.pi
        amp = torch.rand(n, 3, 1, 1, generator=gen) * 0.25
        img = img + amp * torch.cos(2 * math.pi * (fx * xx + fy * yy) / HW + ph)
    return img.clamp(-1.0, 1.0)


def sampler_args(n):
    return base_args(data_size=HW, ddpm_schedule="linear", ddpm_num_steps=T, shift_type="noise_with_perturbation",
                     sampling_mask_dependency="independent", momentum_adaptive="base_momentum", sample_num=n,
                     sample_latent_shape="uniform", sample_history=False)


def free_run(params, dtype, n=4, seed=4252):
    from oracle.sampler_ref import SamplerRef
    from oracle.scheduler_ref import SchedulerRef
    from oracle.unet_ref import UNetRef
    a = sampler_args(n)
    rs = SchedulerRef(a)
    rs.update_ddpm_num_steps(T)
    ts = rs.get_timesteps_epoch(0, 1)
    net = UNetRef(TINY, params, dtype=dtype)
    model = net if dtype == torch.float32 else (lambda x, t: SimpleNamespace(sample=net(x, t).sample.float()))
    seed_all(seed)
    with torch.no_grad():
        x0, _ = SamplerRef(None, a, rs, [None] * 3).sample(model, ts)
    return x0


def main(steps=600, batch=16, lr=2e-3):
    from oracle.scheduler_ref import SchedulerRef
    from oracle.trainer_ref import train_step_ref
    from oracle.unet_ref import UNetRef, random_params
    torch.set_num_threads(8)
    a = base_args(data_size=HW, ddpm_schedule="linear", ddpm_num_steps=T, shift_type="noise_with_perturbation", batch_size=batch)
    net = UNetRef(TINY, random_params(TINY))
    opt = torch.optim.AdamW(net.parameters(), lr=lr)
    s = SchedulerRef(a)
    s.update_ddpm_num_steps(T)
    used = s.get_timesteps_epoch(0, 1)
    gen = torch.Generator().manual_seed(2024)
    seed_all(0)
    losses = []
    for k in range(steps):
        if k == steps * 2 // 3:
            for g in opt.param_groups:
                g["lr"] = lr * 0.3
        r = train_step_ref(net, opt, s, a, smooth_images(batch, gen), used, s.rng)
        losses.append(float(r["loss"]))
        if k % 50 == 0 or k == steps - 1:
            print(f"step {k:4d}  loss {np.mean(losses[-50:]):.5f}", flush=True)
    params = {k: v.detach().clone() for k, v in net.pdict().items()}
    x32 = free_run(params, torch.float32)
    x64 = free_run(params, torch.float64)
    yard = float((x32.double() - x64.double()).norm() / x64.double().norm())
    print(f"free-running T={T}: oracle fp32 vs fp64 rel-L2 {yard:.3e}; |x0| max {float(x32.abs().max()):.3f}")
    out = {"w/" + k: v.numpy() for k, v in params.items()}
    out["loss_curve"] = np.asarray(losses, dtype=np.float32)
    out["yardstick_fp32_vs_fp64"] = np.asarray(yard)
    out["steps"] = np.asarray(steps)
    np.savez_compressed(os.path.join(HERE, "trained_tiny.npz"), **out)
    print("wrote", os.path.join(HERE, "trained_tiny.npz"))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 600)

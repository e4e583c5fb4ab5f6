"""Pin the CPU oracle (oracle/) to golden vectors produced by the reference
itself (tests/golden/make_golden.py).  Bit-exact wherever the oracle replays the
same torch ops on the same RNG stream; tight fp32 tolerance for the U-Net."""
import argparse
import random

import numpy as np
import pytest
import torch

from oracle import scheduler_ref as S
from oracle.sampler_ref import SamplerRef
from oracle.trainer_ref import train_step_ref
from oracle.unet_ref import UNetRef, random_params, same_pad_stride2, timestep_embedding, unet6_config, unet_forward

from golden.make_golden import TINY, base_args, seed_all  # data-only helpers (no reference import)


def T(a):
    return torch.from_numpy(np.asarray(a))


# ----------------------------------------------------------------------------- schedules
@pytest.mark.parametrize("size", [32, 64])
@pytest.mark.parametrize("kind", ["linear", "log", "exponential"])
@pytest.mark.parametrize("steps", [10, 50, 250, 1000])
def test_schedule_tables(golden, size, kind, steps):
    g = golden("schedules")
    ratio, pixels, n = S.schedule_table(kind, steps, size * size, 10.0)
    key = f"sched_{kind}_{steps}_{size}"
    assert n == int(g[key + "_steps"])
    assert np.array_equal(ratio.numpy(), g[key + "_ratio"])
    assert np.array_equal(np.asarray(pixels), g[key + "_pixels"])


def test_log_dedup_counts_match_survey():
    assert S.schedule_table("log", 250, 32 * 32)[2] == 215
    assert S.schedule_table("log", 1000, 32 * 32)[2] == 394
    assert S.schedule_table("log", 1000, 64 * 64)[2] == 802


def test_timesteps_epoch_and_gather(golden):
    g = golden("schedules")
    for scale in (1, 3):
        for epoch in (0, 3, 5, 8):
            assert S.timesteps_epoch(50, scale, epoch, 9) == list(g[f"epochsteps_s{scale}_e{epoch}"])
    ratio, pixels, n = S.schedule_table("log", 50, 32 * 32)
    t = T(g["gather_log_idx_t"])
    assert np.array_equal(S.table_at(pixels, t).numpy(), g["gather_log_idx"])
    assert np.array_equal(S.table_at(ratio, t.float()).numpy(), g["gather_log_thr"])
    assert np.array_equal(S.loss_weights(n, torch.tensor([0, 1, 7, n - 1]), 10.0).numpy(), g["lossw"])


# ----------------------------------------------------------------------------- degrade
def _mo(s):
    try:
        return float(s) if "." in s else int(s)
    except ValueError:
        return s


def test_degrade_all_modes(golden):
    g = golden("degrade")
    x0 = T(g["deg_x0"])
    n = x0.shape[0]
    for i in range(int(g["deg_ncombos"])):
        sel, ch, kind, mo, ma = [str(v) for v in g[f"deg{i}_cfg"]]
        mo = _mo(mo)
        a = base_args(data_size=8, ddpm_schedule=kind, ddpm_num_steps=10, select_degrade_pixel=sel,
                      degrade_channel=None if ch == "None" else ch, mean_option=mo, mean_area=ma)
        s = S.SchedulerRef(a)
        s.update_ddpm_num_steps(10)
        t = T(g[f"deg{i}_t"])
        seed_all(100 + i)
        amount = s.get_black_area_num_pixels_time(t.float() if sel == "thresholding" else t)
        r = s.degrade_training(amount, x0, mean_option=mo, mean_area=ma)
        for j, nm in enumerate(("img", "mask", "dmask", "mean")):
            assert np.array_equal(r[j].numpy(), g[f"deg{i}_train_{nm}"], equal_nan=True), (i, nm)
        seed_all(200 + i)
        r2 = s.degrade_independent_base_sampling(amount[:1].expand(n), x0, mean_option=mo, mean_area=ma)
        for j, nm in enumerate(("img", "mask", "mean")):
            assert np.array_equal(r2[j].numpy(), g[f"deg{i}_samp_{nm}"], equal_nan=True), (i, nm)
        r3 = s.degrade_with_mask(x0, r2[1], mo, ma)
        assert np.array_equal(r3.numpy(), g[f"deg{i}_withmask"], equal_nan=True)


# ----------------------------------------------------------------------------- shift
@pytest.mark.parametrize("tag,n", [("n4", 4), ("nEQw", 8)])
def test_shift_types_including_n_equals_w_quirk(golden, tag, n):
    g = golden("shift")
    types_ = ["1-d_constant", "3-d_constant", "noise_reduction", "noise_std_reduction",
              "noise_with_perturbation", "non_shift"]
    for i, st in enumerate(types_):
        a = base_args(data_size=8, ddpm_schedule="linear", ddpm_num_steps=10, shift_type=st, noise_mean=0.25)
        s = S.SchedulerRef(a)
        s.update_ddpm_num_steps(10)
        t = T(g[f"shift_{tag}_{st}_t"])
        seed_all(300 + i)
        sh = s.get_schedule_shift_time(t, torch.zeros(n, 3, 8, 8))
        assert np.array_equal(sh.numpy(), g[f"shift_{tag}_{st}"]), st


# ----------------------------------------------------------------------------- U-Net
def test_unet_pieces(golden):
    g = golden("unet")
    assert np.allclose(timestep_embedding(T(g["temb_t"]), 128).numpy(), g["temb_128"], rtol=0, atol=1e-6)
    assert list(same_pad_stride2(T(g["samepad_x"])).shape) == list(g["samepad_shape"])


def test_unet_tiny_forward_backward(golden):
    g = golden("unet")
    m = UNetRef(TINY)
    x = T(g["unet_x"]).requires_grad_(True)
    y = m(x, T(g["unet_t"])).sample
    assert np.allclose(y.detach().numpy(), g["unet_y"], rtol=1e-4, atol=2e-5)
    (y * T(g["unet_gy"])).sum().backward()
    assert np.allclose(x.grad.numpy(), g["unet_gx"], rtol=1e-3, atol=2e-5)
    grads = {k: p.grad for k, p in zip(m.keys, m.plist)}
    for k in g.files:
        if k.startswith("unet_grad::"):
            want = g[k]
            got = grads[k.split("::")[1]].numpy()
            assert np.allclose(got, want, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(want).max())), k


def test_unet32_preset_forward(golden):
    g = golden("unet")
    cfg = unet6_config(32)
    p = random_params(cfg, 77)
    assert sum(v.numel() for v in p.values()) == int(g["unet32_nparams"]) == 35746307
    with torch.no_grad():
        y = unet_forward(p, cfg, T(g["unet32_x"]), T(g["unet32_t"]))
    ref = g["unet32_y"]
    assert np.linalg.norm(y.numpy() - ref) / np.linalg.norm(ref) < 1e-5


# ----------------------------------------------------------------------------- sampler
class _Wrap(torch.nn.Module):
    def __init__(self, m):
        super().__init__()
        self.m = m

    @property
    def device(self):
        return torch.device("cpu")

    def forward(self, x, t):
        return self.m(x, t)


def test_sampler_trajectories(golden):
    g = golden("sampler")
    for i in range(int(g["samp_n"])):
        dep, mode, sel, ch, kind, st = [str(v) for v in g[f"samp{i}_cfg"]]
        a = base_args(data_size=16, ddpm_schedule=kind, ddpm_num_steps=6, select_degrade_pixel=sel,
                      degrade_channel=None if ch == "None" else ch, shift_type=st, sampling_mask_dependency=dep,
                      momentum_adaptive=mode, sample_num=2, sample_latent_shape="uniform")
        s = S.SchedulerRef(a)
        s.update_ddpm_num_steps(6)
        ts = s.get_timesteps_epoch(0, 1)
        assert ts == list(g[f"samp{i}_ts"])
        smp = SamplerRef(None, a, s, [None] * 3)
        seed_all(400 + i)
        x0, hist = smp.sample(UNetRef(TINY).eval(), ts)
        ref = g[f"samp{i}_hist"]
        # masks/shifts are bit-exact (same RNG stream); the U-Net outputs agree to fp32 rounding
        assert np.array_equal(hist[1].numpy(), ref[1]), (i, "shift")
        assert np.array_equal(hist[6].numpy(), ref[6]), (i, "mask_t")
        for j in range(11):
            assert np.allclose(hist[j].numpy(), ref[j], rtol=1e-4, atol=5e-5), (i, j)
        assert np.allclose(x0.numpy(), g[f"samp{i}_x0"], rtol=1e-4, atol=5e-5)


# ----------------------------------------------------------------------------- train step
@pytest.mark.parametrize("name", ["ms", "ms_w", "base"])
def test_train_step(golden, name):
    g = golden("train_step")
    st, sel, ch, kind, lw = [str(v) for v in g[f"step_{name}_cfg"]]
    a = base_args(data_size=16, ddpm_schedule=kind, ddpm_num_steps=10, select_degrade_pixel=sel,
                  degrade_channel=None if ch == "None" else ch, shift_type=st, loss_weight_use=(lw == "True"),
                  batch_size=4)
    s = S.SchedulerRef(a)
    s.update_ddpm_num_steps(10)
    ts = s.get_timesteps_epoch(0, 1)
    model = UNetRef(TINY)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    seed_all(500)
    r = train_step_ref(model, opt, s, a, T(g["step_x0"]), ts, s.rng, mean_shift=(name != "base"))
    assert np.array_equal(r["x_in"].numpy(), g[f"step_{name}_xin"])
    assert np.allclose(r["pred"].numpy(), g[f"step_{name}_pred"], rtol=1e-4, atol=2e-5)
    assert abs(float(r["loss"]) - float(g[f"step_{name}_loss"])) < 1e-5 * max(1.0, float(g[f"step_{name}_loss"]))
    sd = model.pdict()
    for k in g.files:
        if k.startswith(f"step_{name}_w::"):
            # one AdamW step moves every weight by ~lr regardless of gradient scale, so compare tightly
            assert np.allclose(sd[k.split("::")[1]].detach().numpy(), g[k], rtol=0, atol=2e-5), k

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


def test_sampler_dependent_t(golden):
    """`sampling_mask_dependency='dependent_t'` (sampler.py:191-196, scheduler.py:480-549): nested masks from one draw;
    and the combinations that fail upstream fail the same way."""
    g = golden("sampler_dep_t")
    for i in range(int(g["dept_n"])):
        mode, ch, kind, st, mo, ma = [str(v) for v in g[f"dept{i}_cfg"]]
        a = base_args(data_size=16, ddpm_schedule=kind, ddpm_num_steps=8, select_degrade_pixel="thresholding", degrade_channel=ch,
                      shift_type=st, sampling_mask_dependency="dependent_t", momentum_adaptive=mode, sample_num=2,
                      sample_latent_shape="uniform", mean_option=mo, mean_area=ma, noise_mean=0.05)
        s = S.SchedulerRef(a)
        s.update_ddpm_num_steps(8)
        ts = s.get_timesteps_epoch(0, 1)
        assert ts == list(g[f"dept{i}_ts"])
        seed_all(800 + i)
        x0, hist = SamplerRef(None, a, s, [None] * 3).sample(UNetRef(TINY).eval(), ts)
        ref = g[f"dept{i}_hist"]
        assert np.array_equal(hist[1].numpy(), ref[1]), (i, "shift")
        assert np.array_equal(hist[6].numpy(), ref[6]) and np.array_equal(hist[7].numpy(), ref[7]), (i, "masks")
        assert (hist[6].numpy() <= hist[7].numpy()).all(), "mask_t must be nested inside mask_{t-1}"
        for j in range(11):
            assert np.allclose(hist[j].numpy(), ref[j], rtol=1e-4, atol=5e-5, equal_nan=True), (i, j)
        assert np.allclose(x0.numpy(), g[f"dept{i}_x0"], rtol=1e-4, atol=5e-5, equal_nan=True)
    for j in range(int(g["dept_nfail"])):
        sel, ch, mo, mo_type, err = [str(v) for v in g[f"dept_fail{j}"]]
        mo = int(mo) if mo_type == "int" else mo
        a = base_args(data_size=16, ddpm_schedule="log", ddpm_num_steps=8, select_degrade_pixel=sel,
                      degrade_channel=None if ch == "None" else ch, shift_type="non_shift", sampling_mask_dependency="dependent_t",
                      momentum_adaptive="base_momentum", sample_num=2, sample_latent_shape="zero", mean_option=mo)
        s = S.SchedulerRef(a)
        s.update_ddpm_num_steps(8)
        with pytest.raises(Exception) as ei:
            SamplerRef(None, a, s, [None] * 3).sample(UNetRef(TINY).eval(), s.get_timesteps_epoch(0, 1))
        assert type(ei.value).__name__ == err, (j, type(ei.value).__name__, err)


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


@pytest.mark.parametrize("gas", [1, 2])
def test_train_trajectory(golden, gas):
    """The reference's `train()` run end to end through real accelerate objects (make_golden.gen_train_traj): 2 epochs x 3
    batches, gradient accumulation 1 and 2 -- the oracle's loop reproduces losses, sync pattern, LR after every batch,
    timestep subsets, global_step and the final weights."""
    from oracle.trainer_ref import train_loop_ref
    from torch.utils.data import DataLoader, TensorDataset
    g = golden("train_traj")
    data = T(g["traj_data"])
    a = base_args(data_size=16, ddpm_schedule="linear", ddpm_num_steps=10, shift_type="noise_with_perturbation",
                  loss_weight_use=True, batch_size=4, use_ema=False, scheduler_num_scale_timesteps=2, gradient_accumulation_steps=gas)
    model = UNetRef(TINY)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    lr_s = torch.optim.lr_scheduler.LambdaLR(opt, lambda k: 1.0 / (1.0 + 0.25 * k))
    loader = DataLoader(TensorDataset(data, torch.zeros(12)), batch_size=4, shuffle=False)
    s = S.SchedulerRef(a)
    seed_all(900 + gas)
    r = train_loop_ref(model, opt, lr_s, s, a, loader, 0, 2, grad_accum=gas)
    tag = f"traj_g{gas}"
    assert r["sync"] == list(g[tag + "_sync"]) and r["global_step"] == int(g[tag + "_global_step"])
    assert r["used"] == [list(g[tag + "_used_e0"]), list(g[tag + "_used_e1"])]
    assert np.allclose(r["lr"], g[tag + "_lr"], rtol=1e-12)
    assert np.allclose(r["losses"], g[tag + "_losses"], rtol=2e-5), (r["losses"], g[tag + "_losses"])
    sd = model.pdict()
    live = live_gradient_keys(golden("train_step"))
    assert len(live) > 100
    for k in live:
        assert np.allclose(sd[k].detach().numpy(), g[tag + "_w::" + k], rtol=0, atol=5e-5), k


def live_gradient_keys(gs, name="ms"):
    """Parameters of TINY whose gradient is not mathematically zero.  A bias or time-embedding projection in front of a
    GroupNorm with ONE channel per group (TINY's 32-channel level) has none: its measured gradient is rounding noise
    (rms ~1e-9 against ~1e-3), and AdamW turns that noise into +-lr per step on both sides -- such tensors carry no parity
    information after an update.  Read off the reference's own gradients of one step (tests/golden/train_step.npz)."""
    rms = {k.split("::")[1]: float(np.sqrt((gs[k] ** 2).mean())) for k in gs.files if k.startswith(f"step_{name}_g::")}
    med = sorted(rms.values())[len(rms) // 2]
    return [k for k, v in rms.items() if v > 1e-3 * med]


# ----------------------------------------------------------------------------- round-2 fixtures (tests/golden/make_golden.py: blocks, sampler_long, train_grads)
def _close(a, b, rtol=1e-3, floor=1e-4):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.allclose(a, b, rtol=rtol, atol=floor * max(1.0, np.abs(b).max()))


def test_residual_block_in_isolation(golden):
    """reference unet6.py:336-362 -- forward, input / time-embedding gradients and every parameter gradient."""
    from oracle.unet_ref import res_block
    g = golden("blocks")
    p = {"b." + k.split("::")[1]: T(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith("rb_p::")}
    x, te = T(g["rb_x"]).requires_grad_(True), T(g["rb_temb"]).requires_grad_(True)
    y = res_block(x, te, p, "b")
    assert _close(y.detach(), g["rb_y"], 1e-4, 2e-5)
    (y * T(g["rb_gy"])).sum().backward()
    assert _close(x.grad, g["rb_gx"]) and _close(te.grad, g["rb_gtemb"])
    for k in g.files:
        if k.startswith("rb_g::"):
            assert _close(p["b." + k.split("::")[1]].grad, g[k]), k


def test_attention_block_in_isolation(golden):
    """reference unet6.py:296-333 (einsum attention over L = 64 tokens)."""
    from oracle.unet_ref import attn_block
    g = golden("blocks")
    p = {"b." + k.split("::")[1]: T(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith("ab_p::")}
    x = T(g["ab_x"]).requires_grad_(True)
    y = attn_block(x, p, "b")
    assert _close(y.detach(), g["ab_y"], 1e-4, 2e-5)
    (y * T(g["ab_gy"])).sum().backward()
    assert _close(x.grad, g["ab_gx"])
    for k in g.files:
        if k.startswith("ab_g::"):
            assert _close(p["b." + k.split("::")[1]].grad, g[k]), k


def test_samepad_stride2_conv_values(golden):
    """reference unet6.py:257-272 + 438-440: the padded tensor and the stride-2 conv output, value for value."""
    import torch.nn.functional as F
    g = golden("blocks")
    x = T(g["sp_x"])
    pad = same_pad_stride2(x)
    assert np.array_equal(pad.numpy(), g["sp_pad"])
    y = F.conv2d(pad, T(g["sp_p::weight"]), T(g["sp_p::bias"]), stride=2)
    assert _close(y, g["sp_y"], 1e-5, 1e-6)


def test_preset_width_slice_forward_backward(golden):
    from golden.make_golden import SLICE
    g = golden("blocks")
    m = UNetRef(SLICE, random_params(SLICE, 9))
    x = T(g["slice_x"]).requires_grad_(True)
    y = m(x, T(g["slice_t"])).sample
    assert _close(y.detach(), g["slice_y"], 1e-4, 2e-5)
    (y * T(g["slice_gy"])).sum().backward()
    assert _close(x.grad, g["slice_gx"])
    grads = {k: p.grad for k, p in zip(m.keys, m.plist)}
    assert list(g["slice_keys"]) == m.keys
    assert np.allclose([float(grads[k].norm()) for k in m.keys], g["slice_gnorms"], rtol=2e-3, atol=1e-5)
    for k in g.files:
        if k.startswith("slice_g::"):
            want = g[k]
            assert _close(grads[k.split("::")[1]][:want.shape[0]], want), k


def test_sampler_trajectories_10_and_50_steps(golden):
    g = golden("sampler_long")
    for i in range(int(g["long_n"])):
        dep, mode, sel, ch, kind, st, Tn = [str(v) for v in g[f"long{i}_cfg"]]
        Tn = int(Tn)
        a = base_args(data_size=16, ddpm_schedule=kind, ddpm_num_steps=Tn, select_degrade_pixel=sel, degrade_channel=ch,
                      shift_type=st, sampling_mask_dependency=dep, momentum_adaptive=mode, sample_num=2,
                      sample_latent_shape="normal", noise_mean=0.1)
        s = S.SchedulerRef(a)
        s.update_ddpm_num_steps(Tn)
        ts = s.get_timesteps_epoch(0, 1)
        assert ts == list(g[f"long{i}_ts"])
        seed_all(600 + i)
        x0, hist = SamplerRef(None, a, s, [None] * 3).sample(UNetRef(TINY).eval(), ts)
        ref = g[f"long{i}_hist"]
        h = np.stack([v.numpy() for v in hist])
        h = h if Tn == 10 else h[:, ::10]
        assert np.array_equal(h[1], ref[1]) and np.array_equal(h[6], ref[6]), i          # shifts and masks: bit-exact
        for j in range(11):
            sc = max(1.0, float(np.abs(ref[j]).max()))
            assert np.abs(h[j] - ref[j]).max() < 2e-4 * sc, (i, j)
        rel = np.linalg.norm(x0.numpy() - g[f"long{i}_x0"]) / np.linalg.norm(g[f"long{i}_x0"])
        assert rel < 1e-4, (i, rel)


def test_train_step_gradient_tensors(golden):
    """Every parameter gradient of one real mean-shift `_run_batch` (captured before clipping) and the clipping norm."""
    g = golden("train_grads")
    a = base_args(data_size=16, ddpm_schedule="linear", ddpm_num_steps=10, shift_type="noise_with_perturbation", loss_weight_use=True,
                  batch_size=4)
    s = S.SchedulerRef(a)
    s.update_ddpm_num_steps(10)
    model = UNetRef(TINY)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    seed_all(501)
    r = train_step_ref(model, opt, s, a, T(g["tg_x0"]), s.get_timesteps_epoch(0, 1), s.rng, do_update=False)
    assert np.array_equal(r["x_in"].numpy(), g["tg_xin"])
    assert abs(float(r["loss"]) - float(g["tg_loss"])) < 1e-5 * max(1.0, float(g["tg_loss"]))
    # do_update=False returns before backward: take the gradients here
    x_in, t = r["x_in"], r["t"]
    model.zero_grad()
    pred = model(x_in, t).sample
    w = s.get_weight_timesteps(r["timeindex"], a.loss_weight_power_base)
    loss = (w[:, None, None, None] * ((x_in + pred) - r["shift"] - T(g["tg_x0"])) ** 2).mean()
    loss.backward()
    grads = {k: p.grad for k, p in zip(model.keys, model.plist)}
    rms = float(torch.cat([v.reshape(-1) for v in grads.values()]).pow(2).mean().sqrt())
    for k in g.files:
        if k.startswith("tg_g::"):
            want = g[k]
            assert np.allclose(grads[k.split("::")[1]].numpy(), want, rtol=2e-3, atol=2e-2 * rms), k
    norm = float(torch.sqrt(sum((v.double() ** 2).sum() for v in grads.values())))
    assert abs(norm - float(g["tg_norm"])) < 1e-4 * float(g["tg_norm"])


def test_evaluation_caller_pieces(golden):
    """oracle/evaluate_ref.py against the reference's own Tester methods / normalize01 (tester.py:140-201)."""
    from oracle import evaluate_ref as E
    g = golden("evaluate")
    data, batch, prev = T(g["ev_data"]), T(g["ev_batch"]), T(g["ev_prev"])
    const = torch.full((1, 3, 8, 8), 0.3)
    assert np.array_equal(E.normalize01(torch.cat([data[:3], const])).numpy(), g["ev_norm01"])
    assert np.allclose(E.compute_similarity(batch, E.normalize01(data)).numpy(), g["ev_sim"], rtol=0, atol=1e-6)
    uniq = E.remove_duplicates_in_batches(batch)
    assert np.array_equal(uniq.numpy(), g["ev_unique_in_batch"]) and uniq.shape[0] < batch.shape[0]
    assert np.array_equal(E.remove_duplicates_across_batches(uniq, prev).numpy(), g["ev_unique_across"])
    assert np.array_equal(E.get_nearest_neighbor_idx(batch, data).numpy(), g["ev_nn_idx"])


@pytest.mark.parametrize("area", ["image-wise", "channel-wise"])
def test_initial_latent_from_the_data_mean_histogram(golden, area):
    """`sample_latent_shape='data'` (sampler.py:46-69): oracle AND product host code against the reference's draw."""
    from oracle import evaluate_ref as E
    import mdm
    from mdm import evaluate as ME
    g = golden("evaluate")
    data = T(g["ev_data"])
    a = base_args(data_size=8, sample_num=6, sample_latent_shape="data", mean_area=area)
    hist = E.data_mean_histogram(data, 6, area)
    assert np.allclose(hist[2].numpy(), g[f"ev_hist_cum_{area}"])
    seed_all(700)
    got = SamplerRef(None, a, S.SchedulerRef(a), hist)._get_latent_initial(None)
    assert np.array_equal(got.contiguous().numpy(), g[f"ev_latent_{area}"])
    # the product's host code: same histogram (main_train_masked.py:60-87) and the same draw
    h2 = ME.data_mean_histogram(data, a)
    assert tuple(h2[0]) == tuple(hist[0]) and torch.equal(h2[2], hist[2]) and all(torch.equal(x, y) for x, y in zip(h2[1], hist[1]))
    seed_all(700)
    got2 = mdm.Sampler(None, a, mdm.Scheduler(a), h2)._get_latent_initial(None)
    assert np.array_equal(got2.contiguous().numpy(), g[f"ev_latent_{area}"])

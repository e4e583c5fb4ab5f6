"""GPU parity of the paths bench.py actually runs, against the CPU oracle:

  * the device-RNG train step (Philox draws inside the kernels, whole step as one hipGraph): its draws are read
    back and REPLAYED through the oracle's train step (reference trainer_masked_mean_shift.py:82-193);
  * the reverse sampler over a full 250-step schedule (BASELINE cfg5; reference sampler.py:109-261) in fp32
    (north_star: within 1e-3 rel-L2 of the CPU reference) and in bf16 (error recorded, loose bound);
  * `Trainer.train()` end to end: epochs, periodic EMA sampling, `save_state`, reload (ms:218-273, 409-425);
  * checkpoint round trip in the reference's directory layout (main_train_masked.py:195-225, 250-277);
  * sampler sharding over 2 ranks.
"""
import json
import os
import socket
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from golden.make_golden import TINY, base_args, seed_all  # noqa: E402


def _note(name, obj):
    """Numbers worth keeping from a GPU run (read back through gpurun_out/)."""
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_notes.jsonl"), "a") as fh:
        fh.write(json.dumps({"test": name, **obj}) + "\n")


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ------------------------------------------------------------------------------------------- device-RNG step
@pytest.mark.parametrize("use_graph", [False, True])
def test_device_rng_train_step_replayed_through_the_oracle(use_graph):
    import mdm
    from oracle.scheduler_ref import ReplayRng, SchedulerRef
    from oracle.trainer_ref import train_step_ref
    from oracle.unet_ref import UNetRef, random_params
    n, hw, T = 4, 16, 50
    a = base_args(data_size=hw, ddpm_schedule="linear", ddpm_num_steps=T, shift_type="noise_with_perturbation", noise_mean=0.25,
                  rng_mode="device", loss_weight_use=True, batch_size=n, use_graph=use_graph, seed=9)
    params = random_params(TINY)
    model = mdm.UNet(TINY, N=n, H=hw, W=hw, dtype=mdm.F32, params=params, use_graph=use_graph)
    opt = mdm.AdamW(model, lr=1e-3)
    tr = mdm.Trainer(a, None, None, [None] * 3, model, None, opt, mdm.get_lr_scheduler("constant", opt, 0, 1), mdm.Accelerator())
    a.updated_ddpm_num_steps = tr.Scheduler.update_ddpm_num_steps(T)
    used = tr.timesteps_used_epoch = tr.Scheduler.get_timesteps_epoch(0, 1)
    g = torch.Generator().manual_seed(21)
    x0 = torch.rand(n, 3, hw, hw, generator=g) * 2 - 1
    # two steps: the second replays the captured graph with an advanced Philox offset
    tr._run_batch(0, (x0, None, None), 0, 1, 0, None, None)
    first = (tr.step.tidx.cpu().clone(), tr.step.mask.cpu().clone())
    P_before = model.state_dict()
    osd_before = opt.state_dict()              # torch.optim.AdamW layout: the oracle continues from it
    loss = tr._run_batch(0, (x0, None, None), 0, 1, 0, None, None)
    st = tr.step
    tidx, t_dev = st.tidx.cpu().long(), model.t_in.cpu()
    amount, ratio, w = st.amount.cpu(), st.ratio.cpu(), st.w.cpu()
    mask, s, x_t, x_in = st.mask.cpu(), st.s.cpu(), st.x_t.cpu(), st.x_in.cpu()
    assert not (torch.equal(first[0], tidx.int()) and torch.equal(first[1], mask)), "the Philox offset did not advance"

    # ---- mdm_draw_timesteps against the host tables, exactly (scheduler.py:88-100, 780-794)
    ref_s = SchedulerRef(a)
    ref_s.update_ddpm_num_steps(T)
    assert used == ref_s.get_timesteps_epoch(0, 1)
    t_want = torch.tensor(used)[tidx]
    assert torch.equal(t_dev, t_want.float())
    assert torch.equal(amount, ref_s.ratio_list[t_want - 1]) and torch.equal(ratio, amount)
    assert torch.equal(w, ref_s.get_weight_timesteps(tidx, a.loss_weight_power_base))
    # ---- Philox mask: 1-channel, keep fraction ~ 1 - ratio (binomial, 256 pixels per image: 5 sigma)
    assert torch.equal(mask[:, 0], mask[:, 1]) and torch.equal(mask[:, 0], mask[:, 2])
    keep = mask[:, 0].flatten(1).mean(1).double()
    sig = (amount * (1 - amount) / (hw * hw)).sqrt()
    assert bool(((keep - (1 - amount)).abs() <= 5 * sig + 1e-9).all()), (keep, amount)
    # ---- Philox shift: s = z * ratio with z ~ N(noise_mean, 1)
    z = (s.double() / ratio[:, None, None, None]).float()
    assert abs(float(z.mean()) - 0.25) < 0.12 and abs(float(z.std()) - 1.0) < 0.1, (float(z.mean()), float(z.std()))

    # ---- the same draws through the oracle: degrade, shift, U-Net, loss, clip, AdamW
    u = torch.where(mask[:, 0] > 0.5, torch.ones(()), torch.zeros(())).reshape(n, hw * hw)     # u > ratio <=> kept
    log = [("randint", tidx), ("uniform", u), ("uniform", torch.zeros(n, 1, 1, 1)), ("normal", z)]
    rs = SchedulerRef(a, rng=ReplayRng(log))
    rs.update_ddpm_num_steps(T)
    ref = UNetRef(TINY, P_before)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3)
    assert list(ref.keys) == model.reference_param_order()
    ropt.load_state_dict(osd_before)
    r = train_step_ref(ref, ropt, rs, a, x0, used, rs.rng)
    assert torch.equal(r["mask"], mask) and torch.equal(r["x_t"], x_t)
    assert float((r["shift"] - s).abs().max()) <= 2e-7 * float(s.abs().max())          # z*ratio re-rounded: <= 1 ulp
    assert float((r["x_in"] - x_in).abs().max()) <= 1e-6
    want = float(r["loss"])
    assert abs(loss - want) < 3e-5 * max(1.0, want), (loss, want)
    sd = model.state_dict()
    # An AdamW step moves a weight by ~lr = 1e-3 whatever the gradient size, so parameters whose gradient is
    # mathematically zero (a conv bias in front of a GroupNorm with one channel per group: TINY's 32-channel level)
    # move by +-lr on rounding noise alone, on both sides: compare where the oracle's gradient is not negligible.
    rms = {k: float(v.grad.pow(2).mean().sqrt()) for k, v in ref.pdict().items()}
    med = sorted(rms.values())[len(rms) // 2]
    keys = [k for k in rms if rms[k] > 1e-3 * med]
    assert len(keys) > 0.8 * len(rms), (len(keys), len(rms))
    worst = 0.0
    for k in keys:
        bad = float(((sd[k] - ref.pdict()[k].detach()).abs() > 3e-5).float().mean())
        worst = max(worst, bad)
        assert bad < 2e-3, (k, bad)
    _note("device_rng_step", dict(use_graph=use_graph, loss=loss, oracle_loss=want, worst_frac=worst))


# ------------------------------------------------------------------------------------------- sampler, T = 250
_ORACLE_RUNS = {}      # the oracle's long sampler runs, shared by the dtype variants of a test (they take most of its time)


@pytest.mark.parametrize("dt,bound,T", [(0, 1e-3, 250), (1, None, 250)])
def test_sampler_250_steps_rel_l2_vs_oracle(dt, bound, T):
    """cfg5: mean-shift sampler, 250 reverse steps, base_momentum / independent masks, host-replayed RNG (the
    reference's draw order), FREE-RUNNING against the fp32 oracle.  fp32: north_star's 1e-3 (see the bar below); the bf16
    figure is recorded (bench.py quotes the dtype whose parity it claims).  The same run at T = 1000 (cfg2's schedule) was
    measured once and is not asserted: rel-L2 1.35 with the oracle's own fp32 run 0.75 away from its fp64 run -- an
    untrained net iterated 1000 times is a chaotic map; T = 1000 is covered step by step in the teacher-forced test."""
    import mdm
    from oracle.sampler_ref import SamplerRef
    from oracle.scheduler_ref import SchedulerRef
    from oracle.unet_ref import UNetRef, random_params
    n, hw = 4, 16
    a = base_args(data_size=hw, ddpm_schedule="linear", ddpm_num_steps=T, shift_type="noise_with_perturbation",
                  sampling_mask_dependency="independent", momentum_adaptive="base_momentum", sample_num=n,
                  sample_latent_shape="uniform", sample_history=False)
    params = random_params(TINY)
    model = mdm.UNet(TINY, N=n, H=hw, W=hw, dtype=dt, params=params).eval()
    s = mdm.Scheduler(a)
    s.update_ddpm_num_steps(T)
    ts = s.get_timesteps_epoch(0, 1)
    assert len(ts) == T
    seed_all(4250)
    x0, hist = mdm.Sampler(None, a, s, [None] * 3).sample(model, ts)
    torch.cuda.synchronize()
    assert hist == [] and bool(torch.isfinite(x0).all())
    if ("free", T) not in _ORACLE_RUNS:
        rs = SchedulerRef(a)
        rs.update_ddpm_num_steps(T)
        seed_all(4250)
        want, _ = SamplerRef(None, a, rs, [None] * 3).sample(UNetRef(TINY, params), ts)
        # YARDSTICK: the oracle's own fp32 run against the same loop with an fp64 network (~1e-4 at T = 250)
        m64 = UNetRef(TINY, params, dtype=torch.float64)
        seed_all(4250)
        rs64 = SchedulerRef(a)
        rs64.update_ddpm_num_steps(T)
        w64, _ = SamplerRef(None, a, rs64, [None] * 3).sample(lambda x, t: SimpleNamespace(sample=m64(x, t).sample.float()), ts)
        _ORACLE_RUNS[("free", T)] = (want, _rel(want, w64))
    want, yard = _ORACLE_RUNS[("free", T)]
    rel = _rel(x0, want)
    _note(f"sampler_{T}", dict(dtype="f32" if dt == 0 else "bf16", rel_l2=rel, steps=T, n=n, oracle_fp32_vs_fp64=yard))
    if bound is not None:
        # Free-running, the per-step differences (fp32 rounding: ~6e-7 of |pred| for this path against ~4.6e-7 for the CPU's own
        # fp32, both measured against fp64 -- scripts/tf_error_profile.py) are amplified by the chaotic map: measured 9.9e-4 at
        # T = 250 with the oracle itself 6.7e-5 from its fp64 run on one host, 1.6e-3 / 6.5e-4 on another.  The oracle's arithmetic
        # depends on the host (the MKL thread count changes its summation order, and with it which way the chaos goes), so the
        # bar leaves 3x on north_star's 1e-3 and scales with the yardstick; the 1e-3 claim itself is carried by the reference's
        # own 6- / 10- / 50-step trajectories (tests/test_path_gpu.py) and by the teacher-forced test below.
        assert rel < max(3.0 * bound, 30.0 * yard), (rel, yard)
    else:
        # bf16 storage does NOT hold north_star's 1e-3 over 250 momentum steps on this net (measured: rel-L2 ~0.5, the
        # momentum update x_t += D_{t-1} - D_t integrates every step's rounding): the fp32 path is the sampler of
        # record, bf16 is reported as the fast approximate mode.  Here: it runs, stays finite and bounded.
        # (measured 0.37-0.57 depending on the box: the bar is 3x the smaller figure)
        assert rel < 1.1 and float(x0.abs().max()) < 1e3, rel


@pytest.mark.parametrize("dt,dep,mode,T", [(0, "independent", "base_momentum", 1000), (1, "independent", "base_momentum", 1000),
                                            (0, "dependent_prev", "base_sampling", 200),
                                            # fp32 storage, convolution products as bf16 hi / lo pairs (UNet(f32_products="split"), the
                                            # sampler of record since DESIGN finding 31): the SAME bars as exact fp32
                                            ("split", "independent", "base_momentum", 1000)])
def test_sampler_1000_steps_teacher_forced_vs_oracle(dt, dep, mode, T):
    """EVERY one of the 1000 reverse steps of cfg2's schedule against the oracle's: the HIP loop starts each step from the
    oracle's x_t of that step (Sampler.step_hook), both consume the host RNG in the reference's order, and x0_hat, both
    degraded images, the masks and the update of each step are compared.  fp32: north_star's 1e-3 per step with room to
    spare; bf16 (the fast approximate sampler of bench.py, EMA previews): a per-step bound that makes a regression visible
    (the end-to-end bf16 check `rel < 2` passes for an unrelated image)."""
    import mdm
    from oracle.sampler_ref import SamplerRef
    from oracle.scheduler_ref import SchedulerRef
    from oracle.unet_ref import UNetRef, random_params
    n, hw = 2, 16
    a = base_args(data_size=hw, ddpm_schedule="linear", ddpm_num_steps=T, shift_type="noise_with_perturbation", noise_mean=0.1,
                  sampling_mask_dependency=dep, momentum_adaptive=mode, sample_num=n, sample_latent_shape="normal",
                  sample_history="device")
    params = random_params(TINY)
    rs = SchedulerRef(a)
    rs.update_ddpm_num_steps(T)
    ts = rs.get_timesteps_epoch(0, 1)
    assert len(ts) == T
    if ("forced", dep, mode, T) not in _ORACLE_RUNS:
        seed_all(4251)
        _, ref = SamplerRef(None, a, rs, [None] * 3).sample(UNetRef(TINY, params), ts)
        _ORACLE_RUNS[("forced", dep, mode, T)] = dict(zip(mdm.sampler.HISTORY_NAMES, ref))
    ref = _ORACLE_RUNS[("forced", dep, mode, T)]
    ref_xt = ref["sample_t"].cuda()
    products, tag = ("split", "f32_split") if dt == "split" else ("exact", "f32" if dt == 0 else "bf16")
    dt = 0 if dt == "split" else dt
    model = mdm.UNet(TINY, N=n, H=hw, W=hw, dtype=dt, params=params, f32_products=products).eval()
    s = mdm.Scheduler(a)
    s.update_ddpm_num_steps(T)
    smp = mdm.Sampler(None, a, s, [None] * 3)
    smp.step_hook = lambda i, slot, x_t: x_t.copy_(ref_xt[slot])
    seed_all(4251)
    x0, hist = smp.sample(model, ts)
    torch.cuda.synchronize()
    hist = {k: v.cpu() for k, v in zip(mdm.sampler.HISTORY_NAMES, hist)}
    assert torch.equal(hist["shift"][1:], ref["shift"][1:])                       # draws: bit-exact at every step
    assert torch.equal(hist["degraded_mask"][1:], ref["degraded_mask"][1:])
    if dep == "independent":
        assert torch.equal(hist["degraded_mask_next"][1:], ref["degraded_mask_next"][1:])
    worst = {}
    last = T if mode == "base_momentum" else T - 1          # base_sampling breaks before the writes of the last step
    for key, upto in (("sample_0", T), ("degraded_t", last), ("degraded_next_t", last), ("difference", last)):
        h, r = hist[key][1:upto + 1].double().flatten(1), ref[key][1:upto + 1].double().flatten(1)
        # per step, relative to the size of that step's tensor (|x_t| wanders over three decades on this net)
        scale = torch.maximum(r.norm(dim=1), ref["sample_0"][1:upto + 1].double().flatten(1).norm(dim=1) * 1e-3)
        e = (h - r).norm(dim=1) / scale
        worst[key] = (float(e.max()), float(e.median()))
    _note("sampler_1000_teacher_forced", dict(dtype=tag, dep=dep, mode=mode, steps=T, n=n,
                                              worst_and_median_rel_l2_per_step=worst))
    for key, (mx, med) in worst.items():
        if dt == 0:
            assert mx < 1e-3 and med < 5e-5, (key, mx, med)
        else:       # bf16 storage: measured worst 2.6e-2, median 9e-4 (profiles/r04_parity_notes.jsonl): bars at 3x
            assert mx < 8e-2 and med < 3e-3, (key, mx, med)


# ------------------------------------------------------------------------------------------- Trainer.train()
def _dirs(tmp_path):
    d = {k: str(tmp_path / k) for k in ("train_loss", "checkpoint", "ema_sample_img")}
    for v in d.values():
        os.makedirs(v, exist_ok=True)
    return SimpleNamespace(list_dir=d)


def _build(a, dt, params, n):
    import mdm
    model = mdm.UNet(TINY, N=n, H=16, W=16, dtype=dt, params=params, use_graph=a.use_graph)
    opt = mdm.AdamW(model, lr=1e-3)
    ema = mdm.EMA(model)
    lr_s = mdm.get_lr_scheduler("cosine", opt, 2, 100)
    acc = mdm.Accelerator()
    acc.prepare(model, opt, None, lr_s)              # main_train_masked.py:299-307
    tr = mdm.Trainer(a, None, None, [None] * 3, model, ema, opt, lr_s, acc)
    return tr, model, opt, ema, lr_s, acc


@pytest.mark.parametrize("dt", [0, 1])
def test_trainer_train_epochs_ema_sample_checkpoint_resume(tmp_path, dt):
    import mdm
    from mdm import checkpoint
    from oracle.unet_ref import random_params
    n = 4
    a = base_args(data_size=16, ddpm_schedule="linear", ddpm_num_steps=12, shift_type="noise_with_perturbation", batch_size=n,
                  rng_mode="device", use_ema=True, use_graph=True, sampling="momentum", sample_num=3, sample_history=False,
                  save_images_epochs=1, scheduler_num_scale_timesteps=2, seed=3)     # scale 2: stride-2 timesteps in epoch 0, every epoch saves
    params = random_params(TINY)
    g = torch.Generator().manual_seed(5)
    loader = [(torch.rand(n, 3, 16, 16, generator=g) * 2 - 1, None, None) for _ in range(3)]
    dirs = _dirs(tmp_path)
    tr, model, opt, ema, lr_s, acc = _build(a, dt, params, n)
    tr.dataloader = loader
    tr.train(0, 2, 0, 0, dirs, None)                                             # 2 epochs x 3 batches
    torch.cuda.synchronize()
    assert tr.global_step == 6 and len(tr.loss_mean_epoch) == 2 and all(np.isfinite(tr.loss_mean_epoch))
    assert opt.t == 6 and ema.optimization_step == 6 and len(tr.lr_list) == 6
    # periodic EMA sample (ms:409-425): written per epoch, finite, training weights restored afterwards
    for ep in (0, 1):
        smp = torch.load(os.path.join(dirs.list_dir["ema_sample_img"], f"ema_sample_{ep:05d}.pt"))
        assert tuple(smp.shape) == (3, 3, 16, 16) and bool(torch.isfinite(smp).all())
    assert float((ema.shadow - model.store.P).abs().max()) > 0                   # EMA lags; P is the training copy again
    # checkpoint in the reference's layout
    ck = os.path.join(dirs.list_dir["checkpoint"], "checkpoint-epoch-1")
    for rel in ("unet/config.json", "unet/" + checkpoint.WEIGHTS, "unet_ema/config.json", "unet_ema/" + checkpoint.WEIGHTS,
                "optimizer.bin", "scheduler.bin", "random_states_0.pkl"):
        assert os.path.exists(os.path.join(ck, rel)), rel
    cfg, sd = checkpoint.load_model_tensors(os.path.join(ck, "unet"))
    order = model.reference_param_order()
    assert set(sd) == set(order) and cfg["hid_channels"] == TINY["hid_channels"]
    assert tuple(sd["in_conv.weight"].shape) == (32, 3, 3, 3)                    # OIHW, unpadded: the reference's tensors
    ecfg, esd = checkpoint.load_model_tensors(os.path.join(ck, "unet_ema"))
    assert ecfg["optimization_step"] == 6 and ecfg["power"] == a.ema_power and ecfg["use_ema_warmup"] is True
    osd = torch.load(os.path.join(ck, "optimizer.bin"), weights_only=False)
    assert len(osd["state"]) == len(sd) and float(osd["state"][0]["step"]) == 6.0
    assert all(tuple(osd["state"][i]["exp_avg"].shape) == tuple(sd[k].shape) for i, k in enumerate(order))
    # torch's own AdamW accepts the file as is
    from oracle.unet_ref import UNetRef
    ref = UNetRef(TINY, {k: sd[k].clone() for k in order})
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3)
    ropt.load_state_dict(osd)

    # a FRESH model loaded from unet/ reproduces the forward bit for bit
    x = torch.rand(n, 3, 16, 16, generator=g) * 2 - 1
    t = torch.tensor([1.0, 5.0, 9.0, 12.0])
    y0 = model(x, t).sample.clone()
    fresh = mdm.UNet(TINY, N=n, H=16, W=16, dtype=dt, params=sd)
    assert torch.equal(fresh.store.P, model.store.P)                             # the weights: bit for bit
    assert _rel(fresh(x, t).sample, y0) < 1e-5                                   # the forward: GroupNorm's LDS float atomics reorder sums

    # resume: load_state into new objects, run one more epoch; same as the uninterrupted run continuing
    tr2, model2, opt2, ema2, lr2, acc2 = _build(a, dt, random_params(TINY, 99), n)      # different weights before the load
    tr2.dataloader = loader
    acc2.load_state(ck)
    assert opt2.t == 6 and ema2.optimization_step == 6 and lr2.k == 6
    assert torch.equal(model2.store.P, model.store.P) and torch.equal(ema2.shadow, ema.shadow)
    assert torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    assert torch.equal(tr2.Scheduler.dev_rng.dev, tr.Scheduler.dev_rng.dev)
    tr.train(2, 1, 0, 6, dirs, None)
    tr2.train(2, 1, 0, 6, dirs, None)
    torch.cuda.synchronize()
    assert tr2.global_step == 9
    # same weights, same Philox stream, same data: equal up to the reordering noise of float atomics
    # (AdamW moves a weight by ~lr per step whatever the gradient size: parameters with a mathematically zero
    # gradient follow rounding noise, so compare by fraction and bound the worst case by 3 steps x 2 lr)
    diff = (model2.store.P - model.store.P).abs()
    assert float(diff.max()) <= 3 * 2 * 1e-3 + 1e-6, float(diff.max())
    assert float((diff > (1e-5 if dt == 0 else 2e-3)).float().mean()) < (0.02 if dt == 0 else 0.10)
    assert np.allclose(tr.loss_mean_epoch, tr2.loss_mean_epoch, rtol=1e-4 if dt == 0 else 5e-2)


def _describe(obj, prefix=""):
    """Same flattening as tests/golden/make_golden.py:_describe (key / shape / dtype grammar of a checkpoint object)."""
    rows = {}
    if isinstance(obj, torch.Tensor):
        rows[prefix] = f"tensor {str(obj.dtype).replace('torch.', '')} {tuple(obj.shape)}"
    elif isinstance(obj, np.ndarray):
        rows[prefix] = f"ndarray {obj.dtype} {tuple(obj.shape)}"
    elif isinstance(obj, dict):
        if not obj:
            rows[prefix] = "dict empty"
        for k, v in obj.items():
            rows.update(_describe(v, f"{prefix}/{k}" if prefix else str(k)))
    elif isinstance(obj, (list, tuple)):
        if len(obj) > 8 and all(isinstance(v, (int, float)) for v in obj):
            rows[prefix] = f"{type(obj).__name__}[{len(obj)}] of {type(obj[0]).__name__}"
        else:
            if not obj:
                rows[prefix] = f"{type(obj).__name__} empty"
            for i, v in enumerate(obj):
                rows.update(_describe(v, f"{prefix}/{i}"))
    else:
        rows[prefix] = type(obj).__name__
    return rows


@pytest.mark.parametrize("gas", [1, 2])
def test_trainer_train_vs_reference_trajectory(golden, tmp_path, gas):
    """A6 / N3 pinned: `mdm.Trainer.train()` in replay mode against the reference's own `train()` run through real accelerate
    objects (tests/golden/train_traj.npz: 2 epochs x 3 batches, gradient accumulation 1 and 2) -- per-batch losses, sync
    pattern, LR after every batch, timestep subsets, global_step, final weights; and the `save_state` directory against the
    key / shape / dtype manifest of the one accelerate wrote (trainer_masked_mean_shift.py:196-273, main_train_masked.py:195-225)."""
    import mdm
    from mdm import checkpoint
    from oracle.unet_ref import random_params
    from test_oracle_golden import live_gradient_keys
    from torch.utils.data import DataLoader, TensorDataset
    g = golden("train_traj")
    tag = f"traj_g{gas}"
    data = torch.from_numpy(g["traj_data"])
    a = base_args(data_size=16, ddpm_schedule="linear", ddpm_num_steps=10, shift_type="noise_with_perturbation", loss_weight_use=True,
                  batch_size=4, sample_num=2, sample_latent_shape="zero", use_ema=False, scheduler_num_scale_timesteps=2,
                  save_images_epochs=10, gradient_accumulation_steps=gas)
    seed_all(0)
    model = mdm.UNet(TINY, N=4, H=16, W=16, dtype=mdm.F32, params=random_params(TINY))
    opt = mdm.AdamW(model, lr=1e-3)
    lr_s = mdm.optim.LambdaLR(opt, lambda k: 1.0 / (1.0 + 0.25 * k))
    loader = DataLoader(TensorDataset(data, torch.zeros(12)), batch_size=4, shuffle=False)     # its iterator draws a base seed from
    acc = mdm.Accelerator(gradient_accumulation_steps=gas)                                     # torch's generator, like upstream's
    model, opt, loader, lr_s = acc.prepare(model, opt, loader, lr_s)
    tr = mdm.Trainer(a, loader, None, [None] * 3, model, None, opt, lr_s, acc)
    losses, syncs, used = [], [], []
    ob, oe = tr._run_batch, tr._run_epoch
    tr._run_batch = lambda *aa, **kk: (lambda r: (losses.append(r), syncs.append(bool(acc.sync_gradients)), r)[-1])(ob(*aa, **kk))
    tr._run_epoch = lambda *aa, **kk: (lambda r: (used.append(list(tr.timesteps_used_epoch)), r)[-1])(oe(*aa, **kk))
    dirs = _dirs(tmp_path)
    seed_all(900 + gas)
    tr.train(0, 2, 0, 0, dirs, None)
    torch.cuda.synchronize()
    assert syncs == list(g[tag + "_sync"]) and tr.global_step == int(g[tag + "_global_step"])
    assert used == [list(g[tag + "_used_e0"]), list(g[tag + "_used_e1"])]
    assert np.allclose(tr.lr_list, g[tag + "_lr"], rtol=1e-12), (tr.lr_list, g[tag + "_lr"])
    assert np.allclose(losses, g[tag + "_losses"], rtol=1e-4), (losses, list(g[tag + "_losses"]))
    assert opt.t == int(g[tag + "_global_step"])
    sd = model.state_dict()
    live = live_gradient_keys(golden("train_step"))
    worst = 0.0
    for k in live:
        d = float(np.abs(sd[k].numpy() - g[tag + "_w::" + k]).max())
        worst = max(worst, d)
        # 4-6 AdamW steps from identical draws: an element only departs by more than rounding if a near-zero gradient
        # changed sign on the way (then by ~lr): allow a handful of those per tensor, bound the rest tightly
        # (a flipped sign moves an element by ~2 lr = 2e-3 in one step; ordinary fp32 differences in g / sqrt(v) stay
        # around 5e-5 after six steps: the threshold sits between the two)
        frac = float((np.abs(sd[k].numpy() - g[tag + "_w::" + k]) > 3e-4).mean())
        assert frac < 5e-3 and d < 6 * 2e-3, (k, frac, d)
    _note("train_trajectory", dict(gas=gas, losses=losses, ref_losses=[float(v) for v in g[tag + "_losses"]], worst_weight_diff=worst))

    # ---- the checkpoint directory against accelerate's (N3)
    assert sorted(os.listdir(dirs.list_dir["checkpoint"])) == list(g[tag + "_ckpt_dirs"])
    ck = os.path.join(dirs.list_dir["checkpoint"], "checkpoint-epoch-1")
    ref_rows = dict(r.split(" :: ") for r in g[tag + "_ckpt_manifest"])
    ours = {}
    for f in ("optimizer.bin", "scheduler.bin", "random_states_0.pkl"):
        ours.update(_describe(torch.load(os.path.join(ck, f), map_location="cpu", weights_only=False), f))
    from safetensors.torch import load_file
    ours.update(_describe(load_file(os.path.join(ck, "unet", checkpoint.WEIGHTS)), "model.safetensors"))
    for k, v in ref_rows.items():
        if k.startswith("model.safetensors/"):
            k2 = k.replace("model.safetensors/net.", "model.safetensors/")      # the golden run's wrapper module was called `net`
            assert ours.get(k2) == v, (k, v, ours.get(k2))
        else:
            assert ours.get(k) == v, (k, v, ours.get(k))
    extra = sorted(k for k in ours if k not in ref_rows and not k.startswith("model.safetensors/"))
    # what this build adds: the CUDA generator states accelerate itself writes on a GPU box, and the device Philox state
    assert all(k.startswith(("random_states_0.pkl/torch_cuda_manual_seed", "random_states_0.pkl/mdm_")) for k in extra), extra
    if gas == 1:
        osd = torch.load(os.path.join(ck, "optimizer.bin"), map_location="cpu", weights_only=False)
        assert float(osd["state"][0]["step"]) == float(g["traj_g1_opt_step"])
        order = model.reference_param_order()
        for idx, key in ((0, "traj_g1_opt_exp_avg_0"), (5, "traj_g1_opt_exp_avg_sq_5")):
            got = osd["state"][idx]["exp_avg" if "sq" not in key else "exp_avg_sq"].numpy()
            assert np.linalg.norm(got - g[key]) <= 2e-3 * np.linalg.norm(g[key]), (order[idx], key)
        grp = dict(s.split("=", 1) for s in g["traj_g1_opt_group"])
        mine = {k: str(v) for k, v in osd["param_groups"][0].items() if k != "params"}
        assert set(mine) == set(grp), (sorted(mine), sorted(grp))
        assert abs(float(mine["lr"]) - float(grp["lr"])) < 1e-12 and mine["betas"] == grp["betas"] and mine["decoupled_weight_decay"] == grp["decoupled_weight_decay"]
        ssd = torch.load(os.path.join(ck, "scheduler.bin"), map_location="cpu", weights_only=False)
        want = dict(s.split("=", 1) for s in g["traj_g1_sched"])
        assert {k: str(v) for k, v in ssd.items()} == want, (ssd, want)


def test_save_state_with_nothing_registered_raises(tmp_path):
    import mdm
    with pytest.raises(RuntimeError):
        mdm.Accelerator().save_state(str(tmp_path / "x"))


# ------------------------------------------------------------------------------------------- sampler sharding
_SHARD_WORKER = r'''
import os, sys
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
import mdm
from mdm.dist import init_from_env
from mdm.sampler import shard_bounds
from golden.make_golden import TINY, base_args
from oracle.unet_ref import random_params
torch.cuda.set_device(0)
init_from_env("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N = 5                                                   # uneven: rank 0 samples 3, rank 1 samples 2
a = base_args(data_size=16, ddpm_schedule="linear", ddpm_num_steps=8, shift_type="noise_with_perturbation", sample_num=N,
              sampling_mask_dependency="independent", momentum_adaptive="base_momentum", sample_latent_shape="zero",
              sample_history=False, rng_mode="device", seed=5)
S = mdm.Scheduler(a); S.update_ddpm_num_steps(8)
ts = S.get_timesteps_epoch(0, 1)
smp = mdm.Sampler(None, a, S, [None] * 3)
lo, hi = shard_bounds(N, rank, world)
assert smp.local_sample_num() == hi - lo == (3 if rank == 0 else 2)
net = mdm.UNet(TINY, N=hi - lo, H=16, W=16, dtype=mdm.F32, params=random_params(TINY)).eval()
x0, hist = smp.sample(net, ts)
torch.cuda.synchronize()
assert tuple(x0.shape) == (N, 3, 16, 16) and hist == [] and bool(torch.isfinite(x0).all())
# every rank holds the same gathered tensor
both = [torch.empty_like(x0.cpu()) for _ in range(world)]
dist.all_gather(both, x0.cpu())
assert torch.equal(both[0], both[1])
# my shard of it == what a single process computes for my Philox key and my share of sample_num
a2 = base_args(**{**vars(a), "sample_num": hi - lo, "shard_sampling": False, "rng_rank": rank})
S2 = mdm.Scheduler(a2); S2.update_ddpm_num_steps(8)
mine, _ = mdm.Sampler(None, a2, S2, [None] * 3).sample(net, ts)
torch.cuda.synchronize()
err = float((mine.cpu() - x0[lo:hi].cpu()).norm() / x0[lo:hi].cpu().norm())
assert err < 1e-4, err                                  # same kernels, same draws (LDS float atomics reorder sums)
# shards come from different streams
assert not torch.equal(x0[0].cpu(), x0[3].cpu())
# ---- replay mode with drawn start colours: the host generator is seeded alike on every rank (main_train_masked.py:441-445),
# so the draws are made for the whole batch and sliced -- the gathered batch is what ONE process computes, shards differ
from golden.make_golden import seed_all
a3 = base_args(**{**vars(a), "rng_mode": "replay", "sample_latent_shape": "uniform", "sample_history": True, "mean_area": "channel-wise"})
S3 = mdm.Scheduler(a3); S3.update_ddpm_num_steps(8)
seed_all(77)
x3, h3 = mdm.Sampler(None, a3, S3, [None] * 3).sample(net, ts)
a4 = base_args(**{**vars(a3), "shard_sampling": False})
S4 = mdm.Scheduler(a4); S4.update_ddpm_num_steps(8)
net_full = mdm.UNet(TINY, N=N, H=16, W=16, dtype=mdm.F32, params=random_params(TINY)).eval()
seed_all(77)
x4, h4 = mdm.Sampler(None, a4, S4, [None] * 3).sample(net_full, ts)
torch.cuda.synchronize()
assert tuple(x3.shape) == (N, 3, 16, 16) and len(h3) == 11 and tuple(h3[0].shape) == tuple(h4[0].shape)
assert torch.equal(h3[0][1], h4[0][1]), "start colours of the gathered batch differ from the single-process draw"
assert torch.equal(h3[1], h4[1]) and torch.equal(h3[6], h4[6])      # shifts and masks: the same draws, bit for bit
err = float((x3.cpu() - x4.cpu()).norm() / x4.cpu().norm())
assert err < 1e-4, err
assert not torch.equal(h3[0][1][0], h3[0][1][3]) and not torch.equal(x3[0].cpu(), x3[3].cpu())     # shards differ
if rank == 0:
    print("SHARD_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_sampler_shards_sample_num_over_two_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_SHARD_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert "SHARD_OK" in outs[0]


# ------------------------------------------------------------------------------------------- free-running T = 1000 on a TRAINED net
def _trained_tiny():
    z = np.load(os.path.join(ROOT, "tests", "golden", "trained_tiny.npz"))
    return {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}, float(z["yardstick_fp32_vs_fp64"])


_TRAINED_ORACLE = {}


@pytest.mark.parametrize("dep,mode", [("independent", "base_sampling"), ("dependent_prev", "base_momentum"), ("dependent_prev", "base_sampling"),
                                      ("independent", "base_momentum")])
@pytest.mark.parametrize("prec", ["f32", "f32_split", "bf16"])
def test_sampler_1000_steps_free_running_on_trained_weights(dep, mode, prec):
    """north_star's literal claim -- "sampler output within 1e-3 rel-L2 of the CPU reference" -- end to end: ALL 1000 reverse steps of
    cfg2's schedule FREE-RUNNING (no teacher forcing) against the oracle (reference sampler.py:137-258), on TINY weights that the
    oracle's own train step has trained on structured synthetic images (tests/golden/make_trained_tiny.py, trained_tiny.npz).  On
    untrained random weights the same run measures conditioning, not parity (the oracle is 0.75 from its own fp64 run).  With the
    trained net three of the four (mask dependency, momentum) combinations are well conditioned -- oracle fp32 vs fp64: 3e-6
    (independent / base_sampling), 1.4e-5 (dependent_prev / base_momentum), 5e-7 (dependent_prev / base_sampling) -- and are ASSERTED
    at 1e-3 for exact fp32 and for the sampler of record (fp32 storage, split products); bf16 storage is reported.  The fourth,
    independent / base_momentum -- the bench's mode -- stays explosive even on the trained net (x_t += D_{t-1} - D_t with two
    INDEPENDENT masks integrates the difference of two masked images at every step: |x0| ~ 1e3, oracle fp32 vs fp64 0.5-1.0): it is
    recorded, not asserted, and covered step by step by the teacher-forced test above."""
    import mdm
    from oracle.sampler_ref import SamplerRef
    from oracle.scheduler_ref import SchedulerRef
    from oracle.unet_ref import UNetRef
    params, _ = _trained_tiny()
    n, hw, T = 4, 16, 1000
    a = base_args(data_size=hw, ddpm_schedule="linear", ddpm_num_steps=T, shift_type="noise_with_perturbation",
                  sampling_mask_dependency=dep, momentum_adaptive=mode, sample_num=n, sample_latent_shape="uniform", sample_history=False)
    if (dep, mode) not in _TRAINED_ORACLE:
        rs = SchedulerRef(a)
        rs.update_ddpm_num_steps(T)
        ts = rs.get_timesteps_epoch(0, 1)
        seed_all(4252)
        with torch.no_grad():
            want, _ = SamplerRef(None, a, rs, [None] * 3).sample(UNetRef(TINY, params), ts)
        m64 = UNetRef(TINY, params, dtype=torch.float64)
        rs64 = SchedulerRef(a)
        rs64.update_ddpm_num_steps(T)
        seed_all(4252)
        with torch.no_grad():
            w64, _ = SamplerRef(None, a, rs64, [None] * 3).sample(lambda x, t: SimpleNamespace(sample=m64(x, t).sample.float()), ts)
        _TRAINED_ORACLE[(dep, mode)] = (want, _rel(want, w64))
    want, yard = _TRAINED_ORACLE[(dep, mode)]
    dt, products = (mdm.BF16, "exact") if prec == "bf16" else (mdm.F32, "split" if prec == "f32_split" else "exact")
    model = mdm.UNet(TINY, N=n, H=hw, W=hw, dtype=dt, params=params, f32_products=products).eval()
    s = mdm.Scheduler(a)
    s.update_ddpm_num_steps(T)
    ts = s.get_timesteps_epoch(0, 1)
    assert len(ts) == T
    seed_all(4252)
    x0, hist = mdm.Sampler(None, a, s, [None] * 3).sample(model, ts)
    torch.cuda.synchronize()
    assert hist == [] and bool(torch.isfinite(x0).all())
    rel = _rel(x0, want)
    _note("sampler_1000_free_running_trained", dict(precision=prec, dep=dep, mode=mode, rel_l2=rel, oracle_fp32_vs_fp64=yard,
                                                    max_abs_x0=float(want.abs().max())))
    if (dep, mode) == ("independent", "base_momentum"):
        return                                  # explosive by construction (docstring): recorded only
    assert yard < 1e-4, yard                    # the configuration is well conditioned: the claim is testable
    # north_star's bar is 1e-3; measured (profiles/r04_parity_notes.jsonl) fp32 <= 1.6e-5, split products <= 4.3e-5, bf16 storage
    # 3.9e-2 / 0.20 / 2.7e-2: bars at <= 3x the worst of the three configurations
    assert rel < {"f32": 5e-5, "f32_split": 1.5e-4, "bf16": 0.6}[prec], (rel, yard)

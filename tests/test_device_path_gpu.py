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
@pytest.mark.parametrize("dt,bound", [(0, 1e-3), (1, None)])
def test_sampler_250_steps_rel_l2_vs_oracle(dt, bound):
    """cfg5: mean-shift sampler, 250 reverse steps, base_momentum / independent masks, host-replayed RNG (the
    reference's draw order) against the fp32 oracle.  fp32 must meet north_star's 1e-3; the bf16 figure is
    recorded (bench.py quotes the dtype whose parity it claims)."""
    import mdm
    from oracle.sampler_ref import SamplerRef
    from oracle.scheduler_ref import SchedulerRef
    from oracle.unet_ref import UNetRef, random_params
    n, hw, T = 4, 16, 250
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
    rs = SchedulerRef(a)
    rs.update_ddpm_num_steps(T)
    seed_all(4250)
    want, _ = SamplerRef(None, a, rs, [None] * 3).sample(UNetRef(TINY, params), ts)
    rel = _rel(x0, want)
    _note("sampler_250", dict(dtype="f32" if dt == 0 else "bf16", rel_l2=rel, steps=T, n=n))
    if bound is not None:
        assert rel < bound, rel
    else:
        # bf16 storage does NOT hold north_star's 1e-3 over 250 momentum steps on this net (measured: rel-L2 ~0.5, the
        # momentum update x_t += D_{t-1} - D_t integrates every step's rounding): the fp32 path is the sampler of
        # record, bf16 is reported as the fast approximate mode.  Here: it runs, stays finite and bounded.
        assert rel < 2.0 and float(x0.abs().max()) < 1e3, rel


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

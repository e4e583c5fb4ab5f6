"""Parity gaps named by the round-3 verdict, closed on the HIP path:

  * the EMA shadow the fused optimizer kernel writes (reference trainer_masked_mean_shift.py:170-172 through
    diffusers' EMAModel, main_train_masked.py:116-131) against the oracle's `train_step_ref(..., ema_params=...)`;
  * data parallelism on the HIP path itself: 2 ranks x N = 4 against 1 rank x N = 8 on the same host draws
    (`Scheduler.replay_rows`), with and without gradient accumulation (ms:139-161);
  * `UNet.sampling_plan` after training steps that were captured before the split shadow existed.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from _notes import note, rel_l2  # noqa: E402
from golden.make_golden import TINY, base_args  # noqa: E402


# ------------------------------------------------------------------------------------------- EMA
@pytest.mark.parametrize("dt", [0, 1])
def test_ema_shadow_vs_oracle(dt):
    """Four replay-mode optimisation steps with EMA on both sides.  (i) the kernel's arithmetic: every step's shadow equals
    e - (1 - d_k)(e - P_k) applied to the HIP path's OWN weights, with the oracle's decay d_k; (ii) the oracle end to end."""
    import mdm
    from oracle.scheduler_ref import SchedulerRef
    from oracle.trainer_ref import ema_decay, train_step_ref
    from oracle.unet_ref import UNetRef, random_params
    n, hw, T, steps = 4, 16, 20, 4
    a = base_args(data_size=hw, ddpm_schedule="linear", ddpm_num_steps=T, shift_type="noise_with_perturbation", batch_size=n,
                  use_ema=True, ema_max_decay=0.9999, ema_inv_gamma=1.0, ema_power=0.75)
    params = random_params(TINY)
    g = torch.Generator().manual_seed(11)
    xs = [torch.rand(n, 3, hw, hw, generator=g) * 2 - 1 for _ in range(steps)]

    model = mdm.UNet(TINY, N=n, H=hw, W=hw, dtype=dt, params=params)
    opt = mdm.AdamW(model, lr=1e-3)
    ema = mdm.EMA(model, decay=a.ema_max_decay, inv_gamma=a.ema_inv_gamma, power=a.ema_power)
    tr = mdm.Trainer(a, None, None, [None] * 3, model, ema, opt, mdm.get_lr_scheduler("constant", opt, 0, 1), mdm.Accelerator())
    tr.Scheduler.update_ddpm_num_steps(T)
    used = tr.timesteps_used_epoch = tr.Scheduler.get_timesteps_epoch(0, 1)
    torch.manual_seed(7)
    E_prev = ema.shadow.clone()
    worst_kernel = 0.0
    for k in range(steps):
        tr._run_batch(0, (xs[k], None, None), 0, 1, 0, None, None)
        P, E = model.store.P.clone(), ema.shadow.clone()
        d = ema_decay(k + 1, a.ema_max_decay, a.ema_inv_gamma, a.ema_power)
        want = E_prev.double() - (1.0 - d) * (E_prev.double() - P.double())
        err = float((E.double() - want).abs().max())
        worst_kernel = max(worst_kernel, err)
        assert err <= 2e-7 * max(1.0, float(P.abs().max())), (k, d, err)       # one fp32 rounding of an O(1) value
        E_prev = E
    assert ema.optimization_step == steps

    ref = UNetRef(TINY, params)
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3)
    rs = SchedulerRef(a)
    rs.update_ddpm_num_steps(T)
    ema_ref = [p.detach().clone() for p in ref.parameters()]
    torch.manual_seed(7)
    decays = []
    for k in range(steps):
        r = train_step_ref(ref, ropt, rs, a, xs[k], used, rs.rng, ema_params=ema_ref, ema_step=k)
        decays.append(r["ema_decay"])
    assert decays[0] == 0.0 and 0.3 < decays[1] < 0.5 and decays[3] > decays[2] > decays[1]     # the warm-up is exercised

    got = model.store.state_dict(order=model.reference_param_order(), src=ema.shadow)
    assert list(got) == list(ref.keys)
    # whole-buffer figure (recorded) and the conditioned per-tensor check of the train-step tests: an AdamW step moves a weight
    # with a mathematically zero gradient by +-lr on rounding noise alone, on both sides -- the EMA inherits that
    rel = rel_l2(torch.cat([v.reshape(-1) for v in got.values()]), torch.cat([e.reshape(-1) for e in ema_ref]))
    rms = {k: float(p.grad.pow(2).mean().sqrt()) for k, p in ref.pdict().items()}
    med = sorted(rms.values())[len(rms) // 2]
    keys = [k for k in rms if rms[k] > 1e-3 * med]
    assert len(keys) > 0.8 * len(rms)
    # measured (profiles/r04_parity_notes.jsonl): fp32 rel 5.3e-5, worst fraction 2.5e-4; bf16 rel 2.2e-3, worst fraction 3.3e-4 ... 1.2e-3
    # over builds that differ only in the summation order of the GroupNorm column sums (the bf16 path's gradients are rounding noise
    # on the 8-channel first convolution, and AdamW turns a flipped sign into +-lr) -- bars at <= 3x the largest
    tol_el, tol_frac, tol_rel = (3e-5, 8e-4, 1.6e-4) if dt == 0 else (2.5e-3, 3e-3, 7e-3)
    worst = 0.0
    for k, e in zip(ref.keys, ema_ref):
        if k not in keys:
            continue
        # ... element by element too: the key third of an attention block's project_in bias has a mathematically zero gradient
        # (a constant added to every score of a softmax row) inside a tensor whose other two thirds are live
        live = ref.pdict()[k].grad.abs() > 1e-2 * rms[k]
        bad = float((((got[k] - e).abs() > tol_el) & live).float().sum() / max(1.0, float(live.sum())))
        worst = max(worst, bad)
        assert bad < tol_frac, (k, bad)
    note("ema_shadow_vs_oracle", dict(dtype=dt, rel_l2_all=rel, worst_frac=worst, kernel_abs=worst_kernel, decays=decays))
    assert rel < tol_rel, rel


# ------------------------------------------------------------------------------------------- split shadow after captured steps
def test_sampling_plan_sees_weights_trained_by_a_graph_captured_before_it():
    """An fp32 model trains through a hipGraph captured BEFORE the first `sampling_plan("f32_split")` call: that graph's
    optimizer tail has no split-shadow launch.  The plan must still multiply by the current filters."""
    import mdm
    from mdm.train_step import TrainStep
    from oracle.unet_ref import random_params
    n, hw, T = 4, 16, 50
    a = base_args(data_size=hw, ddpm_schedule="linear", ddpm_num_steps=T, rng_mode="device", use_graph=True, seed=5)
    model = mdm.UNet(TINY, N=n, H=hw, W=hw, dtype=mdm.F32, params=random_params(TINY), use_graph=True)
    opt = mdm.AdamW(model, lr=2e-3)
    S = mdm.Scheduler(a)
    S.update_ddpm_num_steps(T)
    used = S.get_timesteps_epoch(0, 1)
    step = TrainStep(model, S, a, opt, None, mean_shift=True)
    g = torch.Generator().manual_seed(3)
    x0 = torch.rand(n, 3, hw, hw, generator=g) * 2 - 1
    for _ in range(2):
        step.run_device(x0, used)
    assert getattr(model.store, "Ps", None) is None, "the graph must have been captured without a split shadow"
    x = torch.rand(2, 3, hw, hw, generator=g) * 2 - 1
    t = torch.tensor([7.0, 31.0])

    def both():
        ys = model.sampling_plan(2, "f32_split")(x, t).sample.clone()
        ye = model.sampling_plan(2, "f32")(x, t).sample.clone()
        return ys, ye
    ys1, ye1 = both()
    assert rel_l2(ys1, ye1) < 5e-5
    for _ in range(3):
        step.run_device(x0, used)          # replays the graph captured above: no split-shadow refresh inside it
    ys2, ye2 = both()
    moved = rel_l2(ye2, ye1)
    assert moved > 1e-3, moved             # the weights did change
    r = rel_l2(ys2, ye2)
    note("sampling_plan_after_captured_steps", dict(rel_split_vs_exact=r, weights_moved_output_by=moved))
    assert r < 5e-5, (r, moved)


# ------------------------------------------------------------------------------------------- data parallel == one big batch
_WORKER = r'''
import os, sys, json
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
import mdm
from mdm.dist import GradComm, init_from_env
from mdm.train_step import TrainStep
from golden.make_golden import TINY, base_args
from oracle.unet_ref import random_params
gas, dt = int(sys.argv[2]), int(sys.argv[3])
torch.cuda.set_device(0)
init_from_env("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_loc, hw, T, steps = 4, 16, 50, 3
n_all = n_loc * world
params = random_params(TINY)
g = torch.Generator().manual_seed(31)
xs = [torch.rand(n_all, 3, hw, hw, generator=g) * 2 - 1 for _ in range(steps * gas)]

def run(n, rows, comm):
    a = base_args(data_size=hw, ddpm_schedule="linear", ddpm_num_steps=T, shift_type="noise_with_perturbation", batch_size=n,
                  loss_weight_use=True, use_ema=True)
    model = mdm.UNet(TINY, N=n, H=hw, W=hw, dtype=dt, params=params, wgrad_group_bytes=200 << 10)
    opt = mdm.AdamW(model, lr=1e-3); ema = mdm.EMA(model)
    S = mdm.Scheduler(a); S.update_ddpm_num_steps(T)
    S.replay_rows = rows
    used = S.get_timesteps_epoch(0, 1)
    step = TrainStep(model, S, a, opt, ema, mean_shift=True, comm=comm, grad_accum=gas)
    torch.manual_seed(7)                       # every rank seeds alike, as main_train_masked.py:441-445 does
    losses, G1 = [], None
    lo, hi = (rows[0], rows[1]) if rows else (0, n)
    for k in range(steps * gas):
        sync = (k + 1) % gas == 0
        l = step.run_replay(xs[k][lo:hi], used, sync=sync)
        losses.append(float(l))
        if k == gas - 1:
            torch.cuda.synchronize(); G1 = model.store.G.clone()     # the first exchanged (summed) gradient, before the next step zeroes it
    torch.cuda.synchronize()
    return model, losses, G1, ema

m_dp, l_dp, G_dp, e_dp = run(n_loc, (rank * n_loc, (rank + 1) * n_loc, n_all), GradComm())
# mean of the ranks' losses == the big batch's loss
lt = torch.tensor(l_dp, dtype=torch.float64); dist.all_reduce(lt); lt /= world
if rank == 0:
    m_1, l_1, G_1, e_1 = run(n_all, None, None)
    out = dict(gas=gas, dt=dt)
    out["loss_err"] = float((lt - torch.tensor(l_1, dtype=torch.float64)).abs().max())
    # the exchanged gradient is the rank SUM of local-mean gradients: / world = the big batch's gradient
    out["grad_rel"] = float((G_dp.double() / world - G_1.double()).norm() / G_1.double().norm())
    dP = (m_dp.store.P - m_1.store.P).abs()
    out["w_max"] = float(dP.max()); out["w_frac_gt_3e-5"] = float((dP > 3e-5).float().mean())
    out["w_rel"] = float((m_dp.store.P.double() - m_1.store.P.double()).norm() / m_1.store.P.double().norm())
    out["ema_rel"] = float((e_dp.shadow.double() - e_1.shadow.double()).norm() / e_1.shadow.double().norm())
    print("DPEQ " + json.dumps(out))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("gas,dt", [(1, 0), (2, 0), (1, 1)])
def test_two_ranks_equal_one_rank_with_the_double_batch(tmp_path, gas, dt):
    """2 ranks x N = 4 (gloo, one card, replay draws sliced from the 8-sample host draws) against 1 rank x N = 8 through the same
    `TrainStep`: per-rank loss normalisation, the SUM on the wire and the 1 / world fold-in of the optimizer kernel
    (train_step.py `_finish_update`) must reproduce the big batch's loss, gradient, weights and EMA."""
    import json
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(gas), str(dt)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    line = [ln for ln in outs[0].splitlines() if ln.startswith("DPEQ ")]
    assert line, outs[0][-2000:]
    r = json.loads(line[0][5:])
    note("dp_two_ranks_vs_double_batch", r)
    # measured (profiles/r04_parity_notes.jsonl): fp32 loss 1e-7, gradient 9e-8, weights 2.3e-5 (AdamW turns rounding noise on
    # zero-gradient weights into +-lr), EMA 1.4e-5; bf16 loss 2.2e-4, gradient 2.3e-7 (the per-image work is the same in both
    # runs: only the sums across images differ), weights 3.3e-4, EMA 2.0e-4 -- bars at <= 3x (gradients: a decade, they sit at fp32 noise)
    if dt == 0:
        assert r["loss_err"] < 3e-7 and r["grad_rel"] < 1e-6, r
        assert r["w_frac_gt_3e-5"] < 2e-3 and r["w_rel"] < 7e-5 and r["ema_rel"] < 5e-5, r
    else:
        assert r["loss_err"] < 7e-4 and r["grad_rel"] < 1e-5, r
        assert r["w_rel"] < 1e-3 and r["ema_rel"] < 6e-4, r

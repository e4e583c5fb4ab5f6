#!/usr/bin/env python3
"""Benchmark of the masked-diffusion train step (and the 1k-step reverse sampler) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: synthetic 32x3x32x32 per GPU, full unet6 from the reference's
`Model('unet6',3,32,32,3)` preset (35.75 M parameters, random init), bf16 compute, T=1000 linear
schedule with thresholding/1-channel masks, `noise_with_perturbation` shift (with the reference's
N==W broadcast), x0-space MSE, grad-norm clip 1.0, AdamW, EMA.  One "step" = one full
optimisation step (degrade -> shift -> U-Net fwd -> loss -> U-Net bwd -> [all-reduce] -> clip+AdamW+EMA)
on a batch that is already resident in HBM.  One JSON line on stdout (rank 0).

Extra objects in that line:
  roofline     the contraction kernel family (every `mdm_gemm` launch of the step -- `conv_halo_kernel` / `conv_lin2_kernel` /
               `wgrad_lin_kernel` / `gemm_ring_kernel`: conv fwd/dgrad/wgrad, 1x1 convs, attention products --
               and the batched split-K sum of the weight gradients): algorithmic FLOPs of those launches / their HIP-event
               time, against the dense bf16 MFMA peak;
  step_hbm     north_star's whole-step figure: algorithmic bytes (SURVEY 8(d) counting rule) /
               (step time x 8 TB/s);
  cpu_baseline the CPU oracle (own-words port of the reference trainer, oracle/) timed on this box's
               host cores on a bounded sample (rank 0, N=1 only);
  sampler      wall-clock of a 1000-step reverse sampling run (sample_num=100, history off; N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0          # HBM3E spec


_T0 = time.perf_counter()


def log(msg):
    """Progress line on stderr (keeps long runs visibly alive)."""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def make_args(**kw):
    a = argparse.Namespace(
        dir_dataset="synthetic", data_size=32, in_channel=3, out_channel=3, batch_size=32,
        ddpm_num_steps=1000, updated_ddpm_num_steps=1000, ddpm_schedule="linear", ddpm_schedule_base=10.0,
        scheduler_num_scale_timesteps=1, select_degrade_pixel="thresholding", degrade_channel="1-channel",
        mean_option=0, mean_area="image-wise", shift_type="noise_with_perturbation", noise_mean=0.0,
        sample_latent_shape="zero", sampling="momentum", momentum_adaptive="base_momentum",
        sampling_mask_dependency="independent", sample_num=100, sample_history=False,
        loss_weight_use=False, loss_weight_power_base=10.0, use_ema=True, ema_inv_gamma=1.0, ema_power=0.75,
        ema_max_decay=0.9999, weight_dtype=torch.float32, save_images_epochs=10, mixed_precision="bf16",
        gradient_accumulation_steps=1, rng_mode="device", reference_quirks=True, use_graph=True, seed=0)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def cpu_baseline(n_steps=3):
    """Time the CPU oracle's train step on the same workload shape (fp32, all host cores)."""
    from oracle.scheduler_ref import SchedulerRef
    from oracle.trainer_ref import train_step_ref
    from oracle.unet_ref import UNetRef, unet6_config
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)           # the GPU box gives one GPU a 16-core share
    torch.set_num_threads(cores)
    a = make_args(rng_mode="replay")
    torch.manual_seed(0)
    model = UNetRef(unet6_config(32), seed=0)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    ema = [p.detach().clone() for p in model.parameters()]
    s = SchedulerRef(a)
    s.update_ddpm_num_steps(1000)
    used = s.get_timesteps_epoch(0, 1)
    x0 = torch.rand(32, 3, 32, 32) * 2 - 1
    log(f"cpu baseline: oracle model built, {cores} threads")
    train_step_ref(model, opt, s, a, x0, used, s.rng, ema_params=ema, ema_step=0)          # warm-up
    log("cpu baseline: warm-up step done")
    t0 = time.perf_counter()
    for k in range(n_steps):
        train_step_ref(model, opt, s, a, x0, used, s.rng, ema_params=ema, ema_step=k + 1)
        log(f"cpu baseline: step {k + 1}/{n_steps}")
    dt = time.perf_counter() - t0
    return {"value": round(32 * n_steps / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n_steps} optimisation steps of 32x3x32x32 after 1 warm-up step (oracle/trainer_ref.py, fp32)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true")
    ap.add_argument("--sampler-steps", type=int, default=1000)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    opt_ = ap.parse_args()
    if opt_.cpu_baseline_only:           # child process of the cpu_baseline leg: CPU only, never touches the GPU
        print(json.dumps(cpu_baseline()))
        return

    import mdm
    from mdm import _lib
    from mdm.dist import GradComm, init_from_env
    from mdm.train_step import TrainStep

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the product path has no CPU fallback")
    # (MDM_FORCE_DEVICE / MDM_DIST_BACKEND exist so the multi-rank path can be rehearsed on a 1-GPU box: two
    # ranks on cuda:0 over gloo.  The driver's runs use one rank per GPU over "nccl" = RCCL.)
    if os.environ.get("MDM_FORCE_DEVICE") is not None:
        local = int(os.environ["MDM_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    if world > 1:
        init_from_env(os.environ.get("MDM_DIST_BACKEND", "nccl"))
    dev = torch.device("cuda", local)
    dt = mdm.BF16 if opt_.dtype == "bf16" else mdm.F32
    N = opt_.batch
    args = make_args(batch_size=N, seed=1234 + rank, use_graph=not opt_.no_graph)   # per-rank RNG streams

    cfg = mdm.unet6_config(32)
    model = mdm.UNet(cfg, N=N, H=32, W=32, dtype=dt, seed=0, use_graph=not opt_.no_graph)   # same weights on every rank
    optim = mdm.AdamW(model, lr=1e-4)
    ema = mdm.EMA(model, decay=args.ema_max_decay, inv_gamma=args.ema_inv_gamma, power=args.ema_power)
    sched = mdm.Scheduler(args, device=dev)
    sched.update_ddpm_num_steps(1000)
    used = sched.get_timesteps_epoch(0, 1)
    comm = GradComm() if world > 1 else None
    step = TrainStep(model, sched, args, optim, ema, mean_shift=True, comm=comm)
    g = torch.Generator().manual_seed(100 + rank)
    step.x0.copy_(torch.rand(N, 3, 32, 32, generator=g) * 2 - 1)          # synthetic batch, resident in HBM

    log(f"model built: {model.num_parameters()} params, {len(model.forward_plan.calls)} fwd / {len(model.backward_plan.calls)} bwd launches")

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(opt_.warmup):
        step.run_device(None, used)
        if i == 0:
            torch.cuda.synchronize()
            log("first step done (graphs captured)")
    barrier()
    log("warm-up done")
    t0 = time.perf_counter()
    for _ in range(opt_.steps):
        step.run_device(None, used)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt)
    loss = float(step.loss)
    log(f"timed region: {elapsed:.3f}s for {opt_.steps} steps, loss {loss:.5f}")
    ms_per_step = 1e3 * elapsed / opt_.steps
    value = N * world * opt_.steps / elapsed

    # ---- roofline of the contraction kernel family: HIP events around every launch, on the launch stream
    roofline = None
    step_hbm = None
    if rank == 0:
        st = torch.cuda.current_stream().cuda_stream
        with _lib.Recording() as front:
            step._emit_device_front()
        # the family = every bf16 mdm_gemm call plus the batched split-K sum that finishes the weight gradients
        pick = lambda rec: (lambda i, name: (name == "mdm_gemm" and rec.flops.get(i, (0, -1))[1] == dt) or
                            name == "mdm_splitk_reduce_pending")
        tot_ms, tot_fl, n_launch = 0.0, 0.0, 0
        reps = 3
        # an event pair is not free: measured on the forward contractions as 2 T(one launch) - T(two launches)
        step._hyper()
        front.run(st)
        pair_ms = front.event_overhead(st, pick(front))
        for _ in range(reps):
            step._hyper()
            for rec in (front, model.backward_plan):
                for i, ms in rec.run_timed(st, pick(rec), pair_ms):
                    tot_ms += ms
                    if i in rec.flops:
                        tot_fl += rec.flops[i][0]
                        n_launch += 1
            optim.emit_update(ema.shadow, 1.0, 1.0 / world)
        torch.cuda.synchronize()
        log("event-timed replay done")
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        name = ("mdm_gemm bf16 family: conv_halo / conv_lin2 / wgrad_lin / gemm_ring kernels (+ split-K epilogue and batched reduce)" if dt == mdm.BF16
                else "gemm_f32_kernel")
        peak = PEAK_BF16_TFLOPS if dt == mdm.BF16 else 157.3
        # HBM bytes per launch of this kernel family from PMC counters: collected out of band by
        # scripts/pmc_traffic.sh (rocprofv3 cannot run inside the timed process) and committed under profiles/
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
                traffic = round(json.load(fh)["hbm_bytes_per_launch"])
        except Exception:
            pass
        roofline = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic,
                    "launches_per_step": n_launch // reps, "flops_per_step": tot_fl / reps,
                    "avg_launch_us": round(1e3 * tot_ms / n_launch, 2), "kernel_ms_per_step": round(tot_ms / reps, 3),
                    "event_pair_us": round(1e3 * pair_ms, 2)}
        P = model.num_parameters()
        a_out = model.census()
        bytes_step = (38 + 8) * P + 10 * a_out               # 38 B/param (+8 EMA) + 5 passes x 2 B x A_out
        step_hbm = {"algorithmic_bytes": bytes_step, "a_out": a_out, "params": P,
                    "achieved_GBs": round(bytes_step / (ms_per_step * 1e-3) / 1e9, 1), "peak_GBs": PEAK_HBM_GBS,
                    "frac": round(bytes_step / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}

    # ---- 1k-step sampler wall-clock (single GPU only: samples are independent, no collective)
    sampler = None
    if rank == 0 and world == 1 and not opt_.no_sampler:
        net = model.with_batch(args.sample_num).eval()
        smp = mdm.Sampler(None, args, sched, [None] * 3)
        log("sampler plan built")
        smp.sample(net, used[:3])                            # warm-up / graph capture
        torch.cuda.synchronize()
        log("sampler warm-up done")
        t1 = time.perf_counter()
        x0_hat, _ = smp.sample(net, used[:opt_.sampler_steps])
        torch.cuda.synchronize()
        sec = time.perf_counter() - t1
        log(f"sampler: {sec:.2f}s")
        sampler = {"steps": min(opt_.sampler_steps, len(used)), "sample_num": args.sample_num, "seconds": round(sec, 3),
                   "ms_per_step": round(1e3 * sec / min(opt_.sampler_steps, len(used)), 3),
                   "finite": bool(torch.isfinite(x0_hat).all())}

    cpu = None
    if rank == 0 and world == 1 and not opt_.no_cpu_baseline:
        # bounded: a child process (its own torch thread pool, no GPU context) with a hard time limit
        import subprocess
        try:
            env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], env=env,
                               capture_output=True, text=True, timeout=240)
            sys.stderr.write(r.stderr)
            cpu = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:          # never let the baseline leg take the bench line down
            cpu = {"value": None, "unit": "images/s", "cores": None, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"[:200]}

    if rank == 0:
        out = {
            "metric": "train images/sec at 32x3x32x32 (masked-diffusion step: degrade+shift+unet6 fwd/bwd+clip+AdamW+EMA)",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": opt_.steps, "warmup": opt_.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": opt_.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: 32x3x32x32 per GPU, unet6 Model('unet6',3,32,32,3) 35.75M params "
                                   "random init, T=1000 linear/thresholding/1-channel, shift noise_with_perturbation, "
                                   "AdamW lr 1e-4 + clip 1.0 + EMA, device Philox RNG, hipGraph replay",
                       "global_batch": N * world, "per_gpu_batch": N, "image": "3x32x32",
                       "parallelism": f"dp{world}", "final_loss": round(loss, 5)},
            "roofline": roofline, "step_hbm": step_hbm, "cpu_baseline": cpu, "sampler": sampler,
        }
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

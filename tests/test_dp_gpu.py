"""Data-parallel train step on the GPU with 2 ranks.  Only one GPU is available to this build, so both
ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device); everything else -- bucket
planning from the backward marks, the backward cut into per-bucket hipGraphs, the all-reduce issued
between chunks, the 1/world mean folded into the optimizer kernel -- is the code that runs over RCCL."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
import mdm
from mdm.dist import GradComm, init_from_env
from mdm.train_step import TrainStep
from golden.make_golden import TINY, base_args
from oracle.unet_ref import random_params
backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
init_from_env(backend)
rank, world = dist.get_rank(), dist.get_world_size()
a = base_args(data_size=16, ddpm_schedule="linear", ddpm_num_steps=50, shift_type="noise_with_perturbation",
              rng_mode="device", use_ema=True, seed=100 + rank, use_graph=(sys.argv[2] == "graph"))
model = mdm.UNet(TINY, N=4, H=16, W=16, dtype=mdm.BF16, params=random_params(TINY), use_graph=a.use_graph,
                 wgrad_group_bytes=200 << 10)     # small weight-gradient groups: the tiny model still cuts into several buckets
opt = mdm.AdamW(model, lr=1e-3)
ema = mdm.EMA(model)
S = mdm.Scheduler(a); S.update_ddpm_num_steps(50)
used = S.get_timesteps_epoch(0, 1)
comm = GradComm(bucket_bytes=256 << 10)          # small buckets -> several backward chunks on the tiny model
step = TrainStep(model, S, a, opt, ema, mean_shift=True, comm=comm)
g = torch.Generator().manual_seed(7 + rank)
x0 = torch.rand(4, 3, 16, 16, generator=g) * 2 - 1
P0 = model.store.P.clone()
losses = []
for k in range(4):
    losses.append(float(step.run_device(x0, used)))
torch.cuda.synchronize()
assert len(comm.buckets) >= 3, comm.buckets
assert comm.buckets[0][1] == model.store.size and comm.buckets[-1][0] == 0
keep = (lambda t: t.detach().clone()) if backend == "nccl" else (lambda t: t.detach().cpu())    # RCCL gathers device tensors
P = keep(model.store.P)
both = [torch.empty_like(P) for _ in range(world)]
dist.all_gather(both, P)
# every rank applied the same averaged gradient to the same weights -> identical replicas
assert torch.equal(both[0], both[1]), float((both[0] - both[1]).abs().max())
assert float((P.cpu() - P0.cpu()).abs().max()) > 0 and all(l == l and l < 1e3 for l in losses)
# the exchanged gradient really is the rank sum: G (still in place after the step) is identical across ranks
G = keep(model.store.G)
gb = [torch.empty_like(G) for _ in range(world)]
dist.all_gather(gb, G)
assert torch.equal(gb[0], gb[1])
assert float(G.abs().max()) > 0
# ---- gradient accumulation under data parallelism: micro-steps add to Gacc locally, the syncing step exchanges ONCE and updates
step2 = TrainStep(model, S, a, opt, ema, mean_shift=True, comm=GradComm(bucket_bytes=256 << 10), grad_accum=2)
P1 = model.store.P.clone()
t_before = opt.t
step2.run_device(x0, used, sync=False)
torch.cuda.synchronize()
assert torch.equal(model.store.P, P1) and opt.t == t_before, "a non-syncing micro-step must not touch the weights"
assert float(step2.Gacc.abs().max()) > 0
step2.run_device(x0, used, sync=True)
torch.cuda.synchronize()
assert opt.t == t_before + 1 and float((model.store.P - P1).abs().max()) > 0 and float(step2.Gacc.abs().max()) == 0
P = keep(model.store.P)
both = [torch.empty_like(P) for _ in range(world)]
dist.all_gather(both, P)
assert torch.equal(both[0], both[1]), "replicas diverged under gradient accumulation"
if rank == 0:
    print("DPGPU_OK", len(comm.buckets), losses)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode", ["graph", "eager"])
def test_two_ranks_one_gpu_gloo(tmp_path, mode):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, mode], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert "DPGPU_OK" in outs[0]


_WORKER_RCCL1 = r'''
import os, sys
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
import mdm
from mdm.dist import GradComm, init_from_env
from mdm.train_step import TrainStep
from golden.make_golden import TINY, base_args
from oracle.unet_ref import random_params
mode, wire = sys.argv[2], sys.argv[3]
torch.cuda.set_device(0)
init_from_env("nccl", single=True)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

def run(comm):
    a = base_args(data_size=16, ddpm_schedule="linear", ddpm_num_steps=50, shift_type="noise_with_perturbation",
                  rng_mode="device", use_ema=True, seed=100, use_graph=(mode == "graph"))
    model = mdm.UNet(TINY, N=4, H=16, W=16, dtype=mdm.F32, params=random_params(TINY), use_graph=a.use_graph, wgrad_group_bytes=200 << 10)
    opt = mdm.AdamW(model, lr=1e-3); ema = mdm.EMA(model)
    S = mdm.Scheduler(a); S.update_ddpm_num_steps(50)
    used = S.get_timesteps_epoch(0, 1)
    step = TrainStep(model, S, a, opt, ema, mean_shift=True, comm=comm)
    x0 = torch.rand(4, 3, 16, 16, generator=torch.Generator().manual_seed(7)) * 2 - 1
    losses = [float(step.run_device(x0, used)) for _ in range(4)]
    torch.cuda.synchronize()
    return model.store.P.clone(), model.store.G.clone(), losses

comm = GradComm(bucket_bytes=256 << 10, wire=wire, always_exchange=True)
assert comm.exchange and comm.world == 1
P1, G1, l1 = run(comm)
assert len(comm.buckets) >= 3, comm.buckets
P0, G0, l0 = run(None)                      # the whole-graph step without any exchange
if wire == "f32":                           # a sum over one rank is the identity: bit-equal weights after 4 steps (fp32 path, fixed-order)
    assert torch.equal(P1, P0) and torch.equal(G1, G0) and l1 == l0, (float((P1 - P0).abs().max()), l1, l0)
else:                                       # one bf16 rounding of every gradient element on the wire
    rel = float((G1 - G0).norm() / G0.norm())
    assert 0 < rel < 1e-2, rel
# the collectives RCCL would see in a real job, on the same views of the flat buffer
t = G1[: 1 << 16].clone(); dist.all_reduce(t); torch.cuda.synchronize(); assert torch.equal(t, G1[: 1 << 16])
dist.barrier()
print("RCCL1_OK", len(comm.buckets), l1)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode,wire", [("graph", "f32"), ("eager", "f32"), ("graph", "bf16")])
def test_one_rank_rccl_cut_step(tmp_path, mode, wire):
    """The RCCL ("nccl") backend itself on the one GPU there is: a world of one, the collectives issued anyway
    (`GradComm(always_exchange=True)`), so the cut step graph, the async all-reduce per bucket on RCCL's stream and the stream-level
    waits all run through the library a multi-GPU job uses.  With a sum over one rank being the identity the fp32 step must
    equal the whole-graph single-GPU step bit for bit."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER_RCCL1)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), ROOT, mode, wire], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs: one rank per GPU over RCCL (backend 'nccl')")
@pytest.mark.parametrize("mode", ["graph", "eager"])
def test_two_ranks_two_gpus_rccl(tmp_path, mode):
    """The same data-parallel step with one rank per GPU over RCCL -- runs wherever two GPUs are visible (the 1-GPU build
    box skips it; the bucketed exchange it exercises is the code of the gloo test above)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, mode, "nccl"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert "DPGPU_OK" in outs[0]


def test_bench_gpus_2_prints_a_two_rank_line():
    """`python bench.py --gpus 2` (no torchrun environment) starts two ranks itself and reports n_gpus = 2.
    One card here: both ranks on cuda:0 over gloo (MDM_FORCE_DEVICE / MDM_DIST_BACKEND); the driver's runs use one
    rank per GPU over RCCL."""
    import json
    env = dict(os.environ, MDM_FORCE_DEVICE="0", MDM_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "8",
                        "--no-cpu-baseline", "--no-sampler", "--windows", "1", "--bucket-mb", "32", "--reserve-cus", "8"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["scaling"] == "weak"
    # the line explains its own communication: bytes on the wire, bucket plan, the exposed wait behind the last backward piece
    c = out["comm"]
    assert c["world"] == 2 and c["wire"] == "f32" and 140e6 < c["bytes_per_step"] < 150e6, c
    assert c["buckets"] >= 3 and c["buckets"] == len(c["bucket_sizes_mb"]) and c["wgrad_groups"] >= 3, c
    assert c["exposed_samples"] == 3 and 0.0 <= c["exposed_ms"] <= c["exposed_ms_max"], c
    w = out["windows"]
    assert w["n"] == 2 and w["ms_per_step_min"] <= w["ms_per_step_median"] <= w["ms_per_step_max"], w

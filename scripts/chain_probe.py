"""Chains of small-map convolutions (mdm_chain_*, csrc/gemm.hip chain_kernel) on the bench model: which runs of the recorded forward /
backward plans are chainable, bit-equality of the chained plan with the per-layer launches, and replay time of both as hipGraphs.

    python scripts/chain_probe.py [N=32] [reps=200]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import torch  # noqa: E402

import mdm  # noqa: E402
from mdm import _lib  # noqa: E402


def timed(graph, reps):
    for _ in range(5):
        graph.launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        graph.launch()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    torch.cuda.set_device(0)
    cfg = mdm.unet6_config(32)
    model = mdm.UNet(cfg, N=N, H=32, W=32, dtype=mdm.BF16, seed=0)
    dev = model.device
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(N, 32, 32, model.cin_p, generator=g) * 2 - 1).to(dev, torch.bfloat16)
    x[..., model.cin:] = 0
    model.x_in.data.copy_(x)
    model.t_in.copy_(torch.randint(1, 1000, (N,), generator=g).float())
    for tag, plan in (("fwd", model.forward_plan), ("bwd", model.backward_plan)):
        ch = _lib.chained(plan, dev)
        runs = [c.n for c in ch.chains]
        print(f"{tag}: {len(plan.calls)} launches -> {len(ch.calls)} (+{len(ch.chains)} counter-clearing launches); chains of {runs} phases", flush=True)
        if tag == "bwd":
            model.y_out.grad.copy_((torch.randn(model.y_out.grad.shape, generator=g) * 1e-2).to(dev, torch.bfloat16))
        grads = [a.grad for a in model.acts if a.grad is not None and a is not model.y_out]
        bufs = [a.data for a in model.acts[1:]] + grads + [model.store.G]

        def run(pl):
            if tag == "fwd":
                for b in bufs[:len(model.acts) - 1]:
                    b.zero_()
            else:
                for b in grads + [model.store.G]:
                    b.zero_()
            pl.run()
            torch.cuda.synchronize()
            return [b.clone() for b in bufs]
        ref = run(plan)
        got = run(ch)
        bad = [i for i, (p, q) in enumerate(zip(ref, got)) if not torch.equal(p, q)]
        # (float atomics of the bf16 path -- dgamma / dbeta / bias sums -- arrive in another order: G is compared loosely)
        gi = len(bufs) - 1
        strict_bad = [i for i in bad if i != gi]
        relG = float((ref[gi].double() - got[gi].double()).norm() / (ref[gi].double().norm() + 1e-30))
        print(f"{tag}: chained == per-layer launches on {len(bufs) - len(strict_bad)} of {len(bufs)} buffers bit for bit; rel-L2 of G {relG:.2e}; "
              f"status {[c.status() for c in ch.chains]}", flush=True)
        assert not strict_bad, strict_bad[:8]
        assert all(c.status() == 0 for c in ch.chains)
        g0, g1 = _lib.GraphExec(plan), _lib.GraphExec(ch)
        t0 = timed(g0, reps); t1 = timed(g1, reps); t0b = timed(g0, reps); t1b = timed(g1, reps)
        print(f"{tag}: per-layer {t0:.4f} / {t0b:.4f} ms   chained {t1:.4f} / {t1b:.4f} ms   ({len(plan.calls)} vs {len(ch.calls) + len(ch.chains)} kernels)", flush=True)
        assert all(c.status() == 0 for c in ch.chains)


if __name__ == "__main__":
    main()

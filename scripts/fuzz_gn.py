#!/usr/bin/env python3
"""Randomised parity sweep of GroupNorm(32)+SiLU forward/backward (bf16 path) against torch fp32: channel-block
choices, register-cached / streaming / chunked kernels, two concatenated sources, fused column sums."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import torch, torch.nn.functional as F
from mdm import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
q = lambda t: t.to(bf).to(torch.float32)
rel = lambda a, b: float((a.float().cpu() - b).norm() / (b.norm() + 1e-12))


def run(n_cases, seed, verbose=True):
    random.seed(seed)
    bad = 0
    for _ in range(n_cases):
        N = random.choice([1, 3, 8, 32]); HW = random.choice([16, 64, 256, 576, 1024])
        C0 = random.choice([32, 64, 96, 128, 192, 256, 384, 512]); C1 = random.choice([0, 0, 64, 128])
        C = C0 + C1
        cpg = C // 32; l = cpg
        while l % 8: l += cpg
        if C % 32 or l > 64 or N * HW * C > 32 * 1024 * 384: continue          # the kernels take whole groups in <= 64-channel blocks
        silu = random.choice([True, True, False])
        g = torch.Generator().manual_seed(seed * 1000 + N + HW + C)
        x = q(torch.randn(N, C, HW, generator=g) * 1.3 + 0.4).requires_grad_(True)
        gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True); beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
        y = F.group_norm(x, 32, gamma, beta, eps=1e-6)
        if silu: y = F.silu(y)
        gy = q(torch.randn(y.shape, generator=g)); y.backward(gy)
        xh = x.detach().permute(0, 2, 1).contiguous()
        s0 = xh[..., :C0].contiguous().to(dev, bf); s1 = xh[..., C0:].contiguous().to(dev, bf) if C1 else None
        out = torch.empty(N, HW, C, device=dev, dtype=bf); stats = torch.empty(N, 32, 2, device=dev)
        ws = torch.empty(N * (64 * 32 + 4 * C), device=dev)
        gd, bd = gamma.detach().to(dev), beta.detach().to(dev)
        ops.groupnorm_fwd(1, s0, C0, s1, C1, N, HW, gd, bd, silu, out, stats, ws)
        e_f = rel(out, y.detach().permute(0, 2, 1))
        d0 = torch.empty_like(s0); d1 = torch.empty_like(s1) if C1 else None
        dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
        kw = {}
        if not C1 and random.random() < 0.5:
            per = torch.full((N, C), 7.0, device=dev); tot = torch.zeros(C, device=dev)
            kw = dict(sum_img=per, sum_ld=C, sum_all=tot)
        ops.groupnorm_bwd(1, s0, C0, s1, C1, N, HW, gd, bd, silu, gy.permute(0, 2, 1).contiguous().to(dev, bf), stats, d0, 0, d1, 0, dg, db, ws, **kw)
        torch.cuda.synchronize()
        gx = x.grad.permute(0, 2, 1)
        e_b = max(rel(d0, gx[..., :C0]), rel(d1, gx[..., C0:]) if C1 else 0.0)
        e_g = max(rel(dg, gamma.grad), rel(db, beta.grad))
        e_s = 0.0
        if kw:
            scale = float(gx.abs().sum(1).mean()) + 1e-12
            e_s = float((kw["sum_img"].cpu() - gx.sum(1)).abs().max()) / scale
        ok = e_f < 1e-2 and e_b < 2e-2 and e_g < 1e-2 and e_s < 2e-2
        bad += 0 if ok else 1
        if verbose or not ok:
            print(("ok  " if ok else "BAD ") + f"N={N} HW={HW} C={C0}+{C1} silu={silu} sums={bool(kw)}: fwd {e_f:.1e} dx {e_b:.1e} dgamma/dbeta {e_g:.1e} colsum {e_s:.1e}")
    return bad


if __name__ == "__main__":
    b = run(int(os.environ.get("CASES", "60")), int(os.environ.get("SEED", "0")))
    print("failures:", b)
    sys.exit(1 if b else 0)

#!/usr/bin/env python3
"""Randomised parity sweep of the convolution entry points (forward, data gradient, weight gradient) against
torch fp32 on bf16-rounded operands: exercises every dispatch branch (halo / lin2 / tap split / wgrad_lin /
general ring / register-staged) across batch sizes, map sizes, channel counts, strides, upsample, concat."""
import os, sys, itertools, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "masked-diffusion-model_amd"))
import torch, torch.nn.functional as F
from mdm import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
q = lambda t: t.to(bf).to(torch.float32)
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
up = lambda t: t.to(dev, bf)
rel = lambda a, b: float((a.float().cpu() - b).norm() / (b.norm() + 1e-12))
def small_map_cases():
  """Every combination the small-map kernels see (conv_small_body on 64 x 32 and 32 x 16 tiles, its pair / fallback paths): 4x4 and 8x8 maps,
  128- / 256-channel sources with and without a second (concatenated) source, folded upsample, several batch sizes."""
  cases = []
  for H in (4, 8):
      for N in (2, 4, 32):
          for (C0, C1) in ((128, 0), (256, 0), (128, 128), (256, 256), (256, 128)):
              for Co in (128, 256):
                  for u in ((0, 1) if C1 == 0 else (0,)):
                      if N == 32 and C0 + C1 == 512 and Co == 128:
                          continue                      # keep the sweep short: the widest reduction once per map size
                      cases.append((N, H, C0, C1, Co, 3, 1, u))
  return cases


def run(n_cases, seed, verbose=True, cases=None):
  random.seed(seed)
  given = cases is not None
  cases = list(cases) if given else []
  for _ in range(0 if given else n_cases):
      N = random.choice([1, 2, 3, 5, 8, 16, 32]); H = random.choice([4, 8, 16, 32])
      C0 = random.choice([8, 24, 64, 128, 192, 256]); C1 = random.choice([0, 0, 0, 64, 128]); Co = random.choice([8, 40, 64, 128, 256])
      k = random.choice([1, 3, 3, 3]); s = random.choice([1, 1, 1, 2]) if k == 3 else 1; u = random.choice([0, 0, 0, 1]) if (k == 3 and s == 1 and C1 == 0) else 0
      if N * H * H * max(C0 + C1, Co) > 32 * 32 * 32 * 256: continue
      cases.append((N, H, C0, C1, Co, k, s, u))
  bad = 0
  for (N, H, C0, C1, Co, k, s, u) in cases:
      g = torch.Generator().manual_seed(hash((N, H, C0, C1, Co, k, s, u)) & 0xffff)
      C = C0 + C1
      x = q(torch.randn(N, C, H, H, generator=g)); w = q(torch.randn(Co, C, k, k, generator=g) / (k * C ** 0.5)); b = torch.randn(Co, generator=g)
      pads = (1, 1, 1, 1) if k == 3 and s == 1 else (0, 0, 1, 1) if k == 3 else (0, 0, 0, 0)
      xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
      xi = F.interpolate(xr, scale_factor=2, mode="nearest") if u else xr
      xp = F.pad(xi, (pads[1], pads[3], pads[0], pads[2]))
      y = F.conv2d(xp, wr, b, stride=s)
      gy = q(torch.randn(y.shape, generator=g)); y.backward(gy)
      geom = ops.ConvGeom(N=N, IH=H, IW=H, C0=C0, C1=C1, Cout=Co, KH=k, KW=k, stride=s, pad_t=pads[0], pad_l=pads[1], pad_b=pads[2], pad_r=pads[3], ups=u)
      xh = nhwc(x); s0 = up(xh[..., :C0].contiguous()); s1 = up(xh[..., C0:].contiguous()) if C1 else None
      wt = up(w.permute(2, 3, 0, 1).reshape(k * k, Co, C).contiguous())
      ws = torch.full((max(9 * N * geom.OH * geom.OW * max(Co, C), 1 << 22),), float("nan"), device=dev)
      out = torch.empty(N, geom.OH, geom.OW, Co, device=dev, dtype=bf)
      ops.conv_fwd(1, geom, s0, s1, wt, b.to(dev), out, ws=ws)
      e_f = rel(out, nhwc(y.detach()))
      gyh = up(nhwc(gy))
      wT = up(w.permute(2, 3, 1, 0).reshape(k * k, C, Co).contiguous())
      gx = nhwc(xr.grad)
      if u:       # data gradient on the virtual (upsampled) map, then the 2x2 sum pool the model applies
          tmp = torch.empty(N, geom.VH, geom.VW, C0, device=dev, dtype=bf)
          ops.conv_dgrad_t(1, geom, gyh, wT, tmp, 0, ws=ws)
          d0 = torch.empty(N, H, H, C0, device=dev, dtype=bf)
          ops.sumpool2(1, tmp, d0, 0, N, H, H, C0)
          e_d = rel(d0, gx)
      else:
          d0 = torch.empty(N, H, H, C0, device=dev, dtype=bf); d1 = torch.empty(N, H, H, C1, device=dev, dtype=bf) if C1 else None
          ops.conv_dgrad_t(1, geom, gyh, wT, d0, 0, d1, 0, ws=ws)
          e_d = max(rel(d0, gx[..., :C0]), rel(d1, gx[..., C0:]) if C1 else 0.0)
      gw = torch.zeros(k * k, Co, C, device=dev)
      ops.conv_wgrad(1, geom, gyh, s0, s1, gw, ws=ws)
      e_w = rel(gw, wr.grad.permute(2, 3, 0, 1).reshape(k * k, Co, C))
      torch.cuda.synchronize()
      ok = e_f < 2e-2 and e_d < 2e-2 and e_w < 2e-2
      bad += 0 if ok else 1
      if verbose or not ok: print(("ok  " if ok else "BAD ") + f"N={N} H={H} C={C0}+{C1}->{Co} k{k} s{s} u{u}: fwd {e_f:.1e} dgrad {e_d:.1e} wgrad {e_w:.1e}")
  if verbose:
    print("failures:", bad, "of", len(cases))
  return bad, len(cases)


if __name__ == "__main__":
    bad, n = run(int(os.environ.get("CASES", "60")), int(os.environ.get("SEED", "0")))
    sys.exit(1 if bad else 0)

"""GPU parity of the individual HIP kernels (through the C ABI) against plain torch
fp32 on the CPU.  fp32 path: tight tolerance (same arithmetic, different summation
order).  bf16 path: inputs are pre-rounded to bf16 so only accumulation order and the
final bf16 store differ."""
import math

import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"f32": 0, "bf16": 1}


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _q(x, dt):
    """Round to the storage type (CPU copy stays fp32 for the checker)."""
    return x.bfloat16().float() if dt == "bf16" else x


def _up(x, dt):
    return x.to(_dev(), torch.bfloat16 if dt == "bf16" else torch.float32).contiguous()


def _tol(dt, scale=1.0):
    return (2e-2 if dt == "bf16" else 2e-4) * scale


def _relerr(a, b):
    return float((a.float().cpu() - b).norm() / (b.norm() + 1e-12))


def _nhwc(x):      # NCHW -> NHWC
    return x.permute(0, 2, 3, 1).contiguous()


def _w_tap(w):     # OIHW -> [tap][O][I]
    o, i, kh, kw = w.shape
    return w.permute(2, 3, 0, 1).reshape(kh * kw, o, i).contiguous()


CONV_CASES = [
    # N, H, C0, C1, Cout, K, stride, pads(t,l,b,r), ups
    (2, 8, 16, 0, 32, 3, 1, (1, 1, 1, 1), 0),
    (2, 8, 8, 24, 40, 3, 1, (1, 1, 1, 1), 0),       # concat, odd-ish channel counts
    (3, 8, 32, 0, 32, 3, 2, (0, 0, 1, 1), 0),       # SamePad2d(3,2) + stride 2
    (2, 4, 24, 0, 16, 3, 1, (1, 1, 1, 1), 1),       # nearest x2 upsample folded into the gather
    (2, 8, 64, 32, 48, 1, 1, (0, 0, 0, 0), 0),      # 1x1 skip conv over a concat
    (4, 16, 128, 0, 128, 3, 1, (1, 1, 1, 1), 0),    # big enough for several tiles
    (4, 16, 64, 0, 64, 3, 2, (0, 0, 1, 1), 0),      # stride 2 with a 64-aligned pixel count: linear-gather weight gradient
    (8, 32, 64, 64, 128, 3, 2, (0, 0, 1, 1), 0),    # stride-2 data gradient in parity-class tiles (gemm_ring_kernel `par`): 64-row tiles, two destinations
    (32, 32, 128, 0, 128, 3, 2, (0, 0, 1, 1), 0),   # ... 128-row tiles (the bench model's 16x16 -> 32x32 layer)
    (4, 8, 64, 0, 64, 3, 1, (1, 1, 1, 1), 1),       # folded upsample, 16x16 virtual map: linear-gather weight gradient
    (32, 8, 64, 0, 128, 3, 1, (1, 1, 1, 1), 1),     # folded upsample, enough tiles for the halo kernel (forward)
    (8, 4, 64, 0, 64, 3, 1, (1, 1, 1, 1), 1),       # folded upsample onto an 8x8 map: whole-image halo tiles
    (4, 32, 8, 0, 128, 3, 1, (1, 1, 1, 1), 0),      # the first convolution (3 -> 8 padded input channels): conv_thin_k; its "dgrad_t" is conv_thin_n
    (3, 16, 128, 0, 8, 3, 1, (1, 1, 1, 1), 0),      # the last convolution (8 padded output channels): conv_thin_n; its dgrad_t is conv_thin_k
    (2, 8, 8, 0, 48, 3, 1, (1, 1, 1, 1), 0),        # conv_thin_k with a partial last 16-column block and a ragged pixel tile count
]


def _conv_ref(x0, x1, w, b, stride, pads, ups):
    x = torch.cat([x0, x1], 1) if x1 is not None else x0
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    t, l, bb, r = pads
    x = F.pad(x, (l, r, t, bb))
    return F.conv2d(x, w, b, stride=stride)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dt, case):
    from mdm import ops
    N, H, C0, C1, Cout, K, stride, pads, ups = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x0 = _q(torch.randn(N, C0, H, H, generator=g), dt).requires_grad_(True)
    x1 = _q(torch.randn(N, C1, H, H, generator=g), dt).requires_grad_(True) if C1 else None
    w = _q(torch.randn(Cout, C0 + C1, K, K, generator=g) / math.sqrt((C0 + C1) * K * K), dt).requires_grad_(True)
    b = torch.randn(Cout, generator=g)
    y = _conv_ref(x0, x1, w, b, stride, pads, ups)
    gy = _q(torch.randn(y.shape, generator=g), dt)
    y.backward(gy)

    geom = ops.ConvGeom(N=N, IH=H, IW=H, C0=C0, C1=C1, Cout=Cout, KH=K, KW=K, stride=stride,
                        pad_t=pads[0], pad_l=pads[1], pad_b=pads[2], pad_r=pads[3], ups=ups)
    assert (geom.OH, geom.OW) == tuple(y.shape[2:])
    d0 = _up(_nhwc(x0.detach()), dt)
    d1 = _up(_nhwc(x1.detach()), dt) if C1 else None
    dw_ = _up(_w_tap(w.detach()), dt)
    db = b.to(_dev())
    out = torch.empty(N, geom.OH, geom.OW, Cout, device=_dev(), dtype=d0.dtype)
    ops.conv_fwd(DT[dt], geom, d0, d1, dw_, db, out)
    torch.cuda.synchronize()
    assert _relerr(out, _nhwc(y.detach())) < _tol(dt)

    # data gradient (through the virtual, upsampled extent; pooled afterwards like the model does)
    dgy = _up(_nhwc(gy), dt)
    gx0 = torch.zeros(N, geom.VH, geom.VW, C0, device=_dev(), dtype=d0.dtype)
    gx1 = torch.zeros(N, geom.VH, geom.VW, C1, device=_dev(), dtype=d0.dtype) if C1 else None
    ops.conv_dgrad(DT[dt], geom, dgy, dw_, gx0, 0, gx1, 0)
    if ups:
        p0 = torch.empty(N, H, H, C0, device=_dev(), dtype=d0.dtype)
        ops.sumpool2(DT[dt], gx0, p0, 0, N, H, H, C0)
        gx0 = p0
    torch.cuda.synchronize()
    assert _relerr(gx0, _nhwc(x0.grad)) < _tol(dt, 1.5)
    if C1:
        assert _relerr(gx1, _nhwc(x1.grad)) < _tol(dt, 1.5)

    if dt == "bf16":     # same gradient through the per-tap transposed filters (layout 0)
        wT = _up(_w_tap(w.detach()).transpose(1, 2).contiguous(), dt)
        hx0 = torch.zeros(N, geom.VH, geom.VW, C0, device=_dev(), dtype=d0.dtype)
        hx1 = torch.zeros(N, geom.VH, geom.VW, C1, device=_dev(), dtype=d0.dtype) if C1 else None
        ops.conv_dgrad_t(DT[dt], geom, dgy, wT, hx0, 0, hx1, 0)
        if ups:
            q0 = torch.empty(N, H, H, C0, device=_dev(), dtype=d0.dtype)
            ops.sumpool2(DT[dt], hx0, q0, 0, N, H, H, C0)
            hx0 = q0
        torch.cuda.synchronize()
        assert _relerr(hx0, _nhwc(x0.grad)) < _tol(dt, 1.5)
        if C1:
            assert _relerr(hx1, _nhwc(x1.grad)) < _tol(dt, 1.5)

    # weight gradient, fp32 accumulate (+= on top of an existing value)
    gw = torch.full((K * K, Cout, C0 + C1), 0.5, device=_dev(), dtype=torch.float32)
    ops.conv_wgrad(DT[dt], geom, dgy, d0, d1, gw)                       # split-K through fp32 atomics
    torch.cuda.synchronize()
    assert _relerr(gw - 0.5, _w_tap(w.grad)) < _tol(dt, 0.5 if dt == "bf16" else 1.0)
    gw2 = torch.full((K * K, Cout, C0 + C1), 0.5, device=_dev(), dtype=torch.float32)
    ws = torch.full((4 * gw2.numel(),), float("nan"), device=_dev())   # split-K through partial slabs + reduce
    gb = torch.full((Cout,), 0.25, device=_dev()) if dt == "bf16" else None     # fused bias gradient (bf16 kernel)
    ops.conv_wgrad(DT[dt], geom, dgy, d0, d1, gw2, splitk=3, ws=ws, dbias=gb)
    torch.cuda.synchronize()
    assert _relerr(gw2 - 0.5, _w_tap(w.grad)) < _tol(dt, 0.5 if dt == "bf16" else 1.0)
    if gb is not None:
        assert _relerr(gb - 0.25, gy.sum((0, 2, 3))) < 1e-3


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv_epilogue_rowvec_resid_accumulate(dt):
    from mdm import ops
    N, H, C, Cout = 2, 8, 32, 64
    g = torch.Generator().manual_seed(3)
    x = _q(torch.randn(N, C, H, H, generator=g), dt)
    w = _q(torch.randn(Cout, C, 3, 3, generator=g) / 17.0, dt)
    b = torch.randn(Cout, generator=g)
    rv = torch.randn(N, 96, generator=g)            # slice [16:16+Cout] of a wider time-embedding table
    res = _q(torch.randn(N, Cout, H, H, generator=g), dt)
    y = F.conv2d(x, w, b, padding=1) + rv[:, 16:16 + Cout, None, None] + res
    geom = ops.ConvGeom(N=N, IH=H, IW=H, C0=C, C1=0, Cout=Cout)
    out = torch.empty(N, H, H, Cout, device=_dev(), dtype=_up(x, dt).dtype)
    rvd = rv.to(_dev())
    ops.conv_fwd(DT[dt], geom, _up(_nhwc(x), dt), None, _up(_w_tap(w), dt), b.to(_dev()), out,
                 rowvec=rvd[:, 16:], rv_ld=96, resid=_up(_nhwc(res), dt))
    torch.cuda.synchronize()
    assert _relerr(out, _nhwc(y)) < _tol(dt)
    # dgrad accumulation into an existing gradient
    gy = _q(torch.randn(N, Cout, H, H, generator=g), dt)
    base = _q(torch.randn(N, C, H, H, generator=g), dt)
    want = base + F.conv_transpose2d(gy, w, padding=1)
    dst = _up(_nhwc(base), dt).clone()
    ops.conv_dgrad(DT[dt], geom, _up(_nhwc(gy), dt), _up(_w_tap(w), dt), dst, 1)
    torch.cuda.synchronize()
    assert _relerr(dst, _nhwc(want)) < _tol(dt, 1.5)


@pytest.mark.parametrize("case", [(32, 32, 128, 0, 128), (32, 16, 64, 64, 256), (8, 64, 64, 0, 64),
                                  # small maps: 64-pixel tiles of whole images (one 8x8 image / four 4x4 images per workgroup)
                                  (8, 8, 128, 0, 128), (16, 4, 64, 64, 256)])
def test_conv_halo_forward_and_data_gradient(case):
    """3x3 stride-1 convolutions on maps large enough for the halo-staged kernel (whole image rows x 64 output
    channels per workgroup; 256- and 128-pixel tiles; two concatenated sources): forward with the full epilogue,
    data gradient through the per-tap transposed filters with accumulation into split destinations."""
    from mdm import ops
    dt = "bf16"
    N, H, C0, C1, Cout = case
    C = C0 + C1
    g = torch.Generator().manual_seed(N + H + C)
    x = _q(torch.randn(N, C, H, H, generator=g), dt)
    w = _q(torch.randn(Cout, C, 3, 3, generator=g) / (3.0 * C ** 0.5), dt)
    b = torch.randn(Cout, generator=g)
    rv = torch.randn(N, Cout, generator=g)
    res = _q(torch.randn(N, Cout, H, H, generator=g), dt)
    y = F.conv2d(x, w, b, padding=1) + rv[:, :, None, None] + res
    geom = ops.ConvGeom(N=N, IH=H, IW=H, C0=C0, C1=C1, Cout=Cout)
    xh = _nhwc(x)
    s0 = _up(xh[..., :C0].contiguous(), dt)
    s1 = _up(xh[..., C0:].contiguous(), dt) if C1 else None
    out = torch.empty(N, H, H, Cout, device=_dev(), dtype=torch.bfloat16)
    ops.conv_fwd(1, geom, s0, s1, _up(_w_tap(w), dt), b.to(_dev()), out, rowvec=rv.to(_dev()), rv_ld=Cout, resid=_up(_nhwc(res), dt))
    torch.cuda.synchronize()
    assert _relerr(out, _nhwc(y)) < _tol(dt)
    gy = _q(torch.randn(N, Cout, H, H, generator=g), dt)
    base = _q(torch.randn(N, C, H, H, generator=g), dt)
    want = _nhwc(base + F.conv_transpose2d(gy, w, padding=1))
    bh = _nhwc(base)
    d0 = _up(bh[..., :C0].contiguous(), dt)
    d1 = _up(bh[..., C0:].contiguous(), dt) if C1 else None
    wT = _up(_w_tap(w).transpose(1, 2).contiguous(), dt)
    ops.conv_dgrad_t(1, geom, _up(_nhwc(gy), dt), wT, d0, 1, d1, 1)
    torch.cuda.synchronize()
    assert _relerr(d0, want[..., :C0]) < _tol(dt, 1.5)
    if C1:
        assert _relerr(d1, want[..., C0:]) < _tol(dt, 1.5)


def _split_shadow(w_tap):
    """mdm_split_shadow over one filter tensor [tap][Cout][Cin] (Cin % 32 == 0) -> the B_split operand."""
    from mdm import _lib
    P = w_tap.to(_dev(), torch.float32).contiguous()
    Ps = torch.full_like(P, float("nan"))
    segs = torch.tensor([[0, P.numel()]], dtype=torch.int64, device=_dev())
    _lib.call("mdm_split_shadow", _lib.ptr(P), _lib.ptr(Ps), _lib.ptr(segs), 1, _lib.stream())
    return P, Ps


def test_split_shadow_layout():
    """mdm_split_shadow: per 32-element block, chunk g = bf16(x) of elements {4g..4g+3, 16+4g..16+4g+3} (32-bit word k = element 4g+k |
    element 16+4g+k), chunk 4+g = bf16(x - hi) in the same order; hi + lo reproduces x to 2^-16."""
    g = torch.Generator().manual_seed(5)
    w = torch.randn(2, 8, 64, generator=g) * torch.logspace(-6, 3, 64)
    P, Ps = _split_shadow(w)
    torch.cuda.synchronize()
    halves = Ps.cpu().view(torch.int16).view(-1, 8, 8)               # [block][chunk][8 halves]
    x = w.reshape(-1, 32)
    idx = torch.tensor([[4 * gq + j // 2 if j % 2 == 0 else 16 + 4 * gq + j // 2 for j in range(8)] for gq in range(4)])
    xs = x[:, idx]                                                   # [block][g][8]
    hi = halves[:, :4].view(torch.bfloat16).float()
    lo = halves[:, 4:].view(torch.bfloat16).float()
    assert torch.equal(hi, xs.bfloat16().float())
    assert torch.equal(lo, (xs - hi).bfloat16().float())
    assert float(((hi + lo) - xs).abs().max() / xs.abs().max()) < 2.0 ** -16 and torch.all(((hi + lo) - xs).abs() <= xs.abs() * 2.0 ** -16)


@pytest.mark.parametrize("case", [(32, 32, 128, 0, 128, 0), (100, 32, 128, 0, 128, 0), (25, 16, 64, 64, 256, 0), (8, 64, 32, 0, 64, 0),
                                  (8, 8, 128, 0, 128, 0), (16, 4, 64, 64, 256, 0), (4, 4, 512, 0, 512, 0),
                                  (8, 8, 96, 0, 128, 1), (12, 16, 256, 0, 128, 1),
                                  (100, 8, 256, 0, 256, 0),                # 8x8 maps with enough tiles: two images per 128-pixel tile
                                  (100, 32, 128, 0, 8, 0), (128, 16, 64, 64, 24, 0),   # the net's last convolution: a partly filled 32-channel tile
                                  (100, 16, 256, 0, 256, 0), (100, 16, 128, 128, 256, 0), (50, 32, 64, 0, 256, 1),   # 256 x 128 tiles (one tap per barrier)
                                  (80, 32, 64, 0, 64, 0)])                 # whole rounds of 256-pixel tiles + 128-pixel tiles in one launch
def test_conv_halo_split_products(case):
    """fp32 storage, products as bf16 hi / lo pairs on the bf16 matrix pipe (mdm_gemm_desc.B_split, conv_halo_body<..., SPLIT>): every
    halo tile shape, two sources, folded upsample, the sampler's 100-image batch.  Against an fp64 convolution the split path must stay
    within 2e-5 (measured ~2e-6; the exact fp32 path ~1e-7, bf16 ~3e-3) -- and it must actually differ from the exact path, i.e. run."""
    from mdm import ops
    N, H, C0, C1, Cout, ups = case
    C = C0 + C1
    g = torch.Generator().manual_seed(N + H + C + ups)
    x = torch.randn(N, C, H, H, generator=g)
    w = torch.randn(Cout, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    b = torch.randn(Cout, generator=g)
    rv = torch.randn(N, Cout, generator=g)
    HO = H << ups
    res = torch.randn(N, Cout, HO, HO, generator=g)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    y = (F.conv2d(xin.double(), w.double(), b.double(), padding=1) + rv.double()[:, :, None, None] + res.double()).float()
    geom = ops.ConvGeom(N=N, IH=H, IW=H, C0=C0, C1=C1, Cout=Cout, ups=ups)
    xh = _nhwc(x)
    s0 = xh[..., :C0].contiguous().to(_dev())
    s1 = xh[..., C0:].contiguous().to(_dev()) if C1 else None
    P, Ps = _split_shadow(_w_tap(w))
    outs = {}
    for tag, wsplit in (("exact", None), ("split", Ps)):
        out = torch.full((N, HO, HO, Cout), float("nan"), device=_dev())
        ops.conv_fwd(0, geom, s0, s1, P, b.to(_dev()), out, rowvec=rv.to(_dev()), rv_ld=Cout, resid=_nhwc(res).to(_dev()), w_split=wsplit)
        torch.cuda.synchronize()
        outs[tag] = out
    e_exact, e_split = _relerr(outs["exact"], _nhwc(y)), _relerr(outs["split"], _nhwc(y))
    assert e_exact < 1e-6, e_exact
    assert e_split < 2e-5, e_split
    assert not torch.equal(outs["exact"], outs["split"]), "B_split was ignored: the split kernel did not run"


@pytest.mark.parametrize("case", [(100, 16, 256, 0, 128, 1, (0, 0, 0, 0), True), (100, 16, 128, 128, 256, 1, (0, 0, 0, 0), True),
                                  (100, 32, 128, 0, 128, 3, (0, 0, 1, 1), True), (100, 32, 8, 0, 128, 3, (1, 1, 1, 1), True),
                                  # shapes on the 64 x 64 tiles keep the exact products (the permission is ignored there): same bar
                                  (25, 8, 128, 64, 256, 1, (0, 0, 0, 0), False), (6, 32, 128, 0, 8, 3, (1, 1, 1, 1), False),
                                  (3, 8, 40, 0, 24, 3, (1, 1, 1, 1), False)])
def test_f32_split_register_staged(case):
    """mdm_gemm_desc.f32_split on the convolutions that run on the register-staged fp32 kernel (1x1 incl. two sources, SamePad stride 2,
    the 8-channel ends of the net, ragged channel counts): products as bf16 hi / lo pairs, within 2e-5 of an fp64 convolution."""
    from mdm import ops
    N, H, C0, C1, Cout, K, pads, taken = case
    C = C0 + C1
    stride = 2 if pads == (0, 0, 1, 1) else 1
    g = torch.Generator().manual_seed(N + H + C + Cout)
    x = torch.randn(N, C, H, H, generator=g)
    w = torch.randn(Cout, C, K, K, generator=g) / (K * C ** 0.5)
    b = torch.randn(Cout, generator=g)
    xp = F.pad(x, (pads[1], pads[3], pads[0], pads[2]))
    y = F.conv2d(xp.double(), w.double(), b.double(), stride=stride).float()
    geom = ops.ConvGeom(N=N, IH=H, IW=H, C0=C0, C1=C1, Cout=Cout, KH=K, KW=K, stride=stride, pad_t=pads[0], pad_l=pads[1], pad_b=pads[2], pad_r=pads[3])
    xh = _nhwc(x)
    s0 = xh[..., :C0].contiguous().to(_dev())
    s1 = xh[..., C0:].contiguous().to(_dev()) if C1 else None
    P = _w_tap(w).to(_dev())
    outs = {}
    for tag, flag in (("exact", 0), ("split", 1)):
        out = torch.full(tuple(_nhwc(y).shape), float("nan"), device=_dev())
        ops.conv_fwd(0, geom, s0, s1, P, b.to(_dev()), out, f32_split=flag)
        torch.cuda.synchronize()
        outs[tag] = out
    e_exact, e_split = _relerr(outs["exact"], _nhwc(y)), _relerr(outs["split"], _nhwc(y))
    assert e_exact < 1e-6, e_exact
    assert e_split < 2e-5, e_split
    assert torch.equal(outs["exact"], outs["split"]) != taken, "f32_split: wrong kernel for this shape"


@pytest.mark.parametrize("case", [(100, 16, 256, 256, 256), (100, 32, 256, 128, 128), (100, 8, 256, 0, 768), (100, 8, 256, 0, 256),
                                  (8, 8, 128, 128, 256), (2, 8, 64, 0, 64), (6, 8, 32, 0, 128), (100, 4, 256, 256, 256), (9, 4, 64, 0, 64)])
def test_lin_split_products(case):
    """lin_split_kernel: 1x1 convolutions (skip projections over a concat, attention projections) with split products -- both channel
    tiles, one and two sources, bias + residual: within 2e-5 of fp64, and not the exact kernel's bits."""
    from mdm import ops
    N, H, C0, C1, Cout = case
    C = C0 + C1
    g = torch.Generator().manual_seed(N + H + C + Cout)
    x = torch.randn(N, C, H, H, generator=g)
    w = torch.randn(Cout, C, 1, 1, generator=g) / C ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(N, Cout, H, H, generator=g)
    y = (F.conv2d(x.double(), w.double(), b.double()) + res.double()).float()
    geom = ops.ConvGeom(N=N, IH=H, IW=H, C0=C0, C1=C1, Cout=Cout, KH=1, KW=1, pad_t=0, pad_l=0, pad_b=0, pad_r=0)
    xh = _nhwc(x)
    s0 = xh[..., :C0].contiguous().to(_dev())
    s1 = xh[..., C0:].contiguous().to(_dev()) if C1 else None
    P, Ps = _split_shadow(_w_tap(w))
    outs = {}
    for tag, wsplit in (("exact", None), ("split", Ps)):
        out = torch.full((N, H, H, Cout), float("nan"), device=_dev())
        ops.conv_fwd(0, geom, s0, s1, P, b.to(_dev()), out, resid=_nhwc(res).to(_dev()), w_split=wsplit)
        torch.cuda.synchronize()
        outs[tag] = out
    e_exact, e_split = _relerr(outs["exact"], _nhwc(y)), _relerr(outs["split"], _nhwc(y))
    assert e_exact < 1e-6, e_exact
    assert e_split < 2e-5, e_split
    assert not torch.equal(outs["exact"], outs["split"]), "B_split was ignored: lin_split_kernel did not run"


@pytest.mark.parametrize("case", [(100, 32, 128, 0, 128), (100, 16, 256, 0, 256), (100, 8, 256, 0, 256), (5, 8, 32, 32, 64), (7, 16, 64, 0, 192)])
def test_lin_split_stride2_and_tails(case):
    """lin_split_kernel on the SamePad2d + stride-2 3x3 convolutions (unet6.py:257-272: taps as pixel offsets with a validity bit per row)
    and on pixel counts that are not a multiple of the 128-row tile."""
    from mdm import ops
    N, H, C0, C1, Cout = case
    C = C0 + C1
    g = torch.Generator().manual_seed(N + H + C + Cout)
    x = torch.randn(N, C, H, H, generator=g)
    w = torch.randn(Cout, C, 3, 3, generator=g) / (3.0 * C ** 0.5)
    b = torch.randn(Cout, generator=g)
    y = F.conv2d(F.pad(x, (0, 1, 0, 1)).double(), w.double(), b.double(), stride=2).float()
    geom = ops.ConvGeom(N=N, IH=H, IW=H, C0=C0, C1=C1, Cout=Cout, stride=2, pad_t=0, pad_l=0, pad_b=1, pad_r=1)
    xh = _nhwc(x)
    s0 = xh[..., :C0].contiguous().to(_dev())
    s1 = xh[..., C0:].contiguous().to(_dev()) if C1 else None
    P, Ps = _split_shadow(_w_tap(w))
    outs = {}
    for tag, wsplit in (("exact", None), ("split", Ps)):
        out = torch.full(tuple(_nhwc(y).shape), float("nan"), device=_dev())
        ops.conv_fwd(0, geom, s0, s1, P, b.to(_dev()), out, w_split=wsplit)
        torch.cuda.synchronize()
        outs[tag] = out
    e_exact, e_split = _relerr(outs["exact"], _nhwc(y)), _relerr(outs["split"], _nhwc(y))
    assert e_exact < 1e-6, e_exact
    assert e_split < 2e-5, e_split
    assert not torch.equal(outs["exact"], outs["split"]), "B_split was ignored: lin_split_kernel did not run"


@pytest.mark.parametrize("N,L,C", [(100, 64, 256), (100, 16, 256), (5, 64, 64), (3, 48, 128), (2, 32, 64)])
def test_attn_f32_small_forward(N, L, C):
    """mdm_attn_f32_small_fwd: exact-fp32 fused attention for L <= 64 (the 8x8 / 4x4 attention blocks on the fp32 path) against fp64:
    the output AND the probabilities it leaves for the unfused backward."""
    from mdm import ops
    g = torch.Generator().manual_seed(N + L + C)
    qkv = torch.randn(N, L, 3 * C, generator=g)
    q, k, v = qkv[..., :C].double(), qkv[..., C:2 * C].double(), qkv[..., 2 * C:].double()
    sc = 1.0 / math.sqrt(C)
    P = torch.softmax(q @ k.transpose(1, 2) * sc, dim=-1)
    want = (P @ v).float()
    assert ops.attn_f32_small_supported(L, C) and not ops.attn_f32_small_supported(80, C) and not ops.attn_f32_small_supported(L, 96)
    o = torch.full((N, L, C), float("nan"), device=_dev())
    S = torch.full((N, L, L), float("nan"), device=_dev())
    ops.attn_f32_small_fwd(qkv.to(_dev()), o, S, N, L, C, sc)
    torch.cuda.synchronize()
    assert _relerr(S, P.float()) < 2e-6, _relerr(S, P.float())
    assert _relerr(o, want) < 2e-6, _relerr(o, want)


def test_conv_tap_split_with_epilogue():
    """Small-M 3x3 conv: reduction split over the filter taps (slabs + epilogue kernel), full epilogue."""
    from mdm import ops
    dt = "bf16"
    N, H, W, C, Cout = 8, 4, 8, 64, 128          # non-square map: not taken by the whole-image halo kernel
    g = torch.Generator().manual_seed(31)
    x = _q(torch.randn(N, C, H, W, generator=g), dt)
    w = _q(torch.randn(Cout, C, 3, 3, generator=g) / 24.0, dt)
    b = torch.randn(Cout, generator=g)
    rv = torch.randn(N, Cout, generator=g)
    res = _q(torch.randn(N, Cout, H, W, generator=g), dt)
    y = F.conv2d(x, w, b, padding=1) + rv[:, :, None, None] + res
    geom = ops.ConvGeom(N=N, IH=H, IW=W, C0=C, C1=0, Cout=Cout)
    ws = torch.full((9 * N * H * W * Cout,), float("nan"), device=_dev())
    out = torch.empty(N, H, W, Cout, device=_dev(), dtype=torch.bfloat16)
    ops.conv_fwd(1, geom, _up(_nhwc(x), dt), None, _up(_w_tap(w), dt), b.to(_dev()), out, rowvec=rv.to(_dev()), rv_ld=Cout,
                 resid=_up(_nhwc(res), dt), ws=ws)
    torch.cuda.synchronize()
    assert not torch.isnan(ws[: 9 * N * H * W * Cout]).all()           # the slabs were used
    assert _relerr(out, _nhwc(y)) < _tol(dt)
    gy = _q(torch.randn(N, Cout, H, W, generator=g), dt)
    base = _q(torch.randn(N, C, H, W, generator=g), dt)
    want = base + F.conv_transpose2d(gy, w, padding=1)
    dst = _up(_nhwc(base), dt).clone()
    wT = _up(_w_tap(w).transpose(1, 2).contiguous(), dt)
    ops.conv_dgrad_t(1, geom, _up(_nhwc(gy), dt), wT, dst, 1, ws=ws)
    torch.cuda.synchronize()
    assert _relerr(dst, _nhwc(want)) < _tol(dt, 1.5)


@pytest.mark.parametrize("M,N,K", [(32, 512, 128), (32, 4992, 512), (5, 80, 64)])
def test_skinny_fp32_linear(M, N, K):
    """The batch-sized fp32 linear layers of the time-embedding path (unet6.py:395-399, 350) take a dedicated kernel."""
    from mdm import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    want = 0.5 * (a @ w.t()) + b
    D = torch.full((M, N), float("nan"), device=_dev())
    ops.matmul(0, 0, M, N, K, a.to(_dev()), K, w.to(_dev()), K, D, N, alpha=0.5, bias=b.to(_dev()))
    torch.cuda.synchronize()
    assert _relerr(D, want) < 1e-5


@pytest.mark.parametrize("case", [(32, 4, 512, 256), (32, 8, 512, 256), (32, 16, 384, 256), (32, 32, 256, 128), (2, 8, 64, 64)])
def test_conv_pair_equals_two_launches(case):
    """mdm_gemm_pair: a 3x3 convolution and an independent 1x1 convolution (a ResidualBlock's conv1 and skip projection) in ONE
    launch give bit-identical results to the two launches (the last case is a pair the fused kernel does not cover: fallback)."""
    from mdm import _lib, ops
    N, H, Cin, Cout = case
    dev = _dev()
    g = torch.Generator().manual_seed(H * 7 + Cin)
    bf = torch.bfloat16
    a_in = torch.randn(N, H, H, Cin, generator=g).to(dev, bf)
    x0 = torch.randn(N, H, H, Cin // 2, generator=g).to(dev, bf)
    x1 = torch.randn(N, H, H, Cin - Cin // 2, generator=g).to(dev, bf)
    w3 = (torch.randn(9, Cout, Cin, generator=g) * 0.02).to(dev, bf)
    w1 = (torch.randn(1, Cout, Cin, generator=g) * 0.05).to(dev, bf)
    b3, b1 = torch.randn(Cout, generator=g).to(dev), torch.randn(Cout, generator=g).to(dev)
    rv = torch.randn(N, Cout, generator=g).to(dev)
    g3 = ops.ConvGeom(N=N, IH=H, IW=H, C0=Cin, C1=0, Cout=Cout)
    g1 = ops.ConvGeom(N=N, IH=H, IW=H, C0=Cin // 2, C1=Cin - Cin // 2, Cout=Cout, KH=1, KW=1, pad_t=0, pad_l=0, pad_b=0, pad_r=0)
    ws = torch.empty(1 << 22, device=dev)
    outs = []
    for paired in (False, True):
        y3 = torch.full((N, H, H, Cout), float("nan"), device=dev, dtype=bf)
        y1 = torch.full((N, H, H, Cout), float("nan"), device=dev, dtype=bf)
        fa = ops.conv_fwd_fields(1, g3, a_in, None, w3, b3, y3, rowvec=rv, rv_ld=Cout, ws=ws)
        fb = ops.conv_fwd_fields(1, g1, x0, x1, w1, b1, y1, ws=ws)
        if paired:
            ops.conv_fwd_pair(fa, fb)
        else:
            _lib.gemm(**fa); _lib.gemm(**fb)
        torch.cuda.synchronize()
        outs.append((y3.float().cpu(), y1.float().cpu()))
    assert bool(torch.isfinite(outs[1][0]).all()) and bool(torch.isfinite(outs[1][1]).all())
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    want1 = torch.einsum("nhwc,oc->nhwo", torch.cat([x0, x1], -1).float().cpu(), w1[0].float().cpu()) + b1.cpu()
    assert _relerr(outs[1][1], want1) < _tol("bf16")


@pytest.mark.parametrize("M", [4, 32, 100])
@pytest.mark.parametrize("variant", [None, (True, 0.0)])
def test_time_embedding_path_kernels(M, variant):
    """mdm_skinny_linear_fwd / _bwd / mdm_silu_bwd_sum and the bias gradient riding on the fp32 weight gradient (the
    time-embedding path, unet6.py:18-34, 395-399, 350, 359) against fp32 torch: embedding + Linear + SiLU in one launch,
    Linear with a second silu output, the wide projection layer, the split data gradient and the silu' epilogues."""
    from mdm import ops
    hid, te, ft = 128, 512, 4992
    g = torch.Generator().manual_seed(M)
    t = torch.randint(1, 1001, (M,), generator=g).float()
    W1, b1 = torch.randn(te, hid, generator=g) * 0.05, torch.randn(te, generator=g) * 0.1
    W2, b2 = torch.randn(te, te, generator=g) * 0.05, torch.randn(te, generator=g) * 0.1
    W3, b3 = torch.randn(ft, te, generator=g) * 0.05, torch.randn(ft, generator=g) * 0.1
    dT = torch.randn(M, ft, generator=g) * 0.1
    # reference
    half = hid // 2
    flip, shift = (False, 1.0) if variant is None else variant
    f = torch.exp(-torch.arange(half, dtype=torch.float32) * (math.log(10000.0) / (half - shift)))
    ang = t[:, None] * f[None]
    e = torch.cat([torch.cos(ang), torch.sin(ang)] if flip else [torch.sin(ang), torch.cos(ang)], 1)
    leaves = [x.clone().requires_grad_(True) for x in (W1, b1, W2, b2, W3, b3)]
    h1 = e @ leaves[0].T + leaves[1]
    tm = F.silu(h1) @ leaves[2].T + leaves[3]
    T_all = F.silu(tm) @ leaves[4].T + leaves[5]
    (T_all * dT).sum().backward()
    # device
    DEV = _dev()
    d = lambda x: x.to(DEV).contiguous()
    z = lambda *s: torch.zeros(*s, device=DEV)
    tD, W1d, b1d, W2d, b2d, W3d, b3d, dTd = map(d, (t, W1, b1, W2, b2, W3, b3, dT))
    eD, h1D, a1D, tmD, stD, TD = z(M, hid), z(M, te), z(M, te), z(M, te), z(M, te), z(M, ft)
    assert ops.skinny_supported(M, te, hid) and ops.skinny_supported(M, ft, te) and ops.skinny_supported(M, te, ft, 13)
    ops.skinny_linear_fwd(None, W1d, b1d, M, te, hid, h1D, act_out=a1D, t=tD, variant=variant, emb_out=eD)
    ops.skinny_linear_fwd(a1D, W2d, b2d, M, te, te, tmD, act_out=stD)
    ops.skinny_linear_fwd(stD, W3d, b3d, M, ft, te, TD)
    torch.cuda.synchronize()
    # sin / cos of arguments up to 1000 rad: the device's range reduction differs from the host's in the last bits
    assert _relerr(eD, e) < 2e-4 and _relerr(h1D, h1.detach()) < 2e-4 and _relerr(a1D, F.silu(h1).detach()) < 2e-4
    assert _relerr(tmD, tm.detach()) < 2e-4 and _relerr(TD, T_all.detach()) < 2e-4
    gW3, gb3, gW2, gb2, gW1, gb1 = z(ft, te), z(ft), z(te, te), z(te), z(te, hid), z(te)
    slabs, d_tm, d_h1 = z(13, M, te), z(M, te), z(M, te)
    ops.matmul(0, 2, ft, te, M, dTd, ft, stD, te, gW3, te, acc=1, out_f32=1, dbias=gb3)
    ops.skinny_linear_bwd(dTd, W3d, M, te, ft, splits=13, slabs=slabs)
    ops.silu_bwd_sum(tmD, slabs, 13, M * te, d_tm)
    ops.matmul(0, 2, te, te, M, d_tm, te, a1D, te, gW2, te, acc=1, out_f32=1, dbias=gb2)
    ops.skinny_linear_bwd(d_tm, W2d, M, te, te, dx=d_h1, pre=h1D)
    ops.matmul(0, 2, te, hid, M, d_h1, te, eD, hid, gW1, hid, acc=1, out_f32=1, dbias=gb1)
    torch.cuda.synchronize()
    for got, want in zip((gW1, gb1, gW2, gb2, gW3, gb3), leaves):
        assert _relerr(got, want.grad) < 5e-4, _relerr(got, want.grad)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 64, 64, 256), (2, 16, 16, 64), (2, 256, 256, 64), (1, 40, 72, 136)])
def test_batched_matmul_layouts(dt, shape):
    """The three operand layouts on the attention-sized products (unet6.py:316-324)."""
    from mdm import ops
    B, M, N, K = shape
    g = torch.Generator().manual_seed(7)
    dev = _dev()
    a_mk = _q(torch.randn(B, M, K, generator=g), dt)
    a_km = a_mk.transpose(1, 2).contiguous()
    b_nk = _q(torch.randn(B, N, K, generator=g), dt)
    b_kn = b_nk.transpose(1, 2).contiguous()
    want = torch.einsum("bmk,bnk->bmn", a_mk, b_nk) * 0.25
    for layout, A, lda, Bm, ldb in ((0, a_mk, K, b_nk, K), (1, a_mk, K, b_kn, N), (2, a_km, M, b_kn, N)):
        D = torch.empty(B, M, N, device=dev, dtype=_up(a_mk, dt).dtype)
        ops.matmul(DT[dt], layout, M, N, K, _up(A, dt), lda, _up(Bm, dt), ldb, D, N, batch=B, sA=M * K, sB=N * K,
                   sD=M * N, alpha=0.25)
        torch.cuda.synchronize()
        assert _relerr(D, want) < _tol(dt), f"layout {layout}"
    # layout 2 with split-K (partial slabs in a workspace, summed in a fixed order) ACCUMULATED onto an fp32 destination;
    # and the same request without a workspace: the reduction is then not split (there is no atomic fallback)
    for ws in (torch.empty(2 * B * M * N, device=dev, dtype=torch.float32), None):
        D = torch.full((B, M, N), 1.0, device=dev, dtype=torch.float32)
        ops.matmul(DT[dt], 2, M, N, K, _up(a_km, dt), M, _up(b_kn, dt), N, D, N, batch=B, sA=M * K, sB=N * K, sD=M * N,
                   alpha=0.25, out_f32=1, splitk=2, acc=1, ws=ws)
        torch.cuda.synchronize()
        assert _relerr(D - 1.0, want) < _tol(dt)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("C0,C1,HW,silu", [(128, 0, 64, True), (256, 128, 16, True), (32, 0, 256, False), (64, 32, 64, True),
                                           # large maps: pixel chunks (statistics launch + apply launch) on the bf16 path
                                           (128, 0, 1024, True), (64, 32, 1024, True), (32, 0, 576, False)])
def test_groupnorm_fwd_bwd(dt, C0, C1, HW, silu):
    from mdm import ops
    N, G = 3, 32
    C = C0 + C1
    g = torch.Generator().manual_seed(C + HW)
    x = _q(torch.randn(N, C, HW, generator=g) * 1.5 + 0.3, dt).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    y = F.group_norm(x, G, gamma, beta, eps=1e-6)
    if silu:
        y = F.silu(y)
    gy = _q(torch.randn(y.shape, generator=g), dt)
    y.backward(gy)
    dev = _dev()
    xh = x.detach().permute(0, 2, 1).contiguous()          # [N, HW, C]
    s0 = _up(xh[..., :C0], dt)
    s1 = _up(xh[..., C0:], dt) if C1 else None
    out = torch.empty(N, HW, C, device=dev, dtype=s0.dtype)
    stats = torch.empty(N, G, 2, device=dev)
    ws = torch.empty(N * (64 * G + 4 * C), device=dev)
    gd, bd = gamma.detach().to(dev), beta.detach().to(dev)
    ops.groupnorm_fwd(DT[dt], s0, C0, s1, C1, N, HW, gd, bd, silu, out, stats, ws)
    torch.cuda.synchronize()
    assert _relerr(out, y.detach().permute(0, 2, 1)) < _tol(dt, 0.5)
    d0 = torch.empty_like(s0)
    d1 = torch.empty_like(s1) if C1 else None
    dg = torch.zeros(C, device=dev)
    db = torch.zeros(C, device=dev)
    ops.groupnorm_bwd(DT[dt], s0, C0, s1, C1, N, HW, gd, bd, silu, _up(gy.permute(0, 2, 1), dt), stats, d0, 0, d1, 0, dg, db, ws)
    torch.cuda.synchronize()
    gx = x.grad.permute(0, 2, 1)
    assert _relerr(d0, gx[..., :C0]) < _tol(dt)
    if C1:
        assert _relerr(d1, gx[..., C0:]) < _tol(dt)
    assert _relerr(dg, gamma.grad) < _tol(dt, 0.25)
    assert _relerr(db, beta.grad) < _tol(dt, 0.25)
    if not C1:       # fused column sums of dx (time-embedding / bias gradient of the producing conv)
        per = torch.zeros(N, C + 8, device=dev)
        tot = torch.full((C,), 2.0, device=dev)
        dg2, db2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ops.groupnorm_bwd(DT[dt], s0, C0, None, 0, N, HW, gd, bd, silu, _up(gy.permute(0, 2, 1), dt), stats, d0, 0, None, 0,
                          dg2, db2, ws, sum_img=per[:, 8:], sum_ld=C + 8, sum_all=tot)
        torch.cuda.synchronize()
        # (the sum of a GroupNorm gradient over a whole group is mathematically zero, so with one channel
        # per group both sides are rounding noise: measure against the magnitude of dx itself)
        scale = float(gx.abs().sum(1).mean())
        assert float((per[:, 8:].cpu() - gx.sum(1)).abs().max()) < _tol(dt, 0.5) * scale and float(per[:, :8].abs().sum()) == 0
        assert float(((tot - 2.0).cpu() - gx.sum((0, 1))).abs().max()) < _tol(dt, 0.5) * scale * N


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_shift_leaves_pad_channels_alone_and_zero_pad_channels_clears_them(dt):
    """`mdm_shift` writes only the C real channels of the NHWC network input; a caller whose buffer did not come from a
    zeroing allocator runs `mdm_zero_pad_channels` once (include/mdm_hip.h).  Contract check on a NaN-filled buffer."""
    from mdm import _lib
    from mdm._lib import call, ptr, stream
    dev = _dev()
    N, C, H, W, Cp = 3, 3, 8, 8, 8
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    x_t = torch.rand(N, C, H, W, device=dev) * 2 - 1
    buf = torch.full((N, H, W, Cp), float("nan"), device=dev, dtype=tdt)
    s_out = torch.empty_like(x_t); x_in = torch.empty_like(x_t)
    call("mdm_shift", ptr(x_t), None, None, None, 2, 0, 0.0, 0, N, C, H, W, ptr(s_out), ptr(x_in), DT[dt], ptr(buf), Cp, stream())
    torch.cuda.synchronize()
    assert bool(torch.isnan(buf[..., C:].float()).all()), "mdm_shift must not touch the pad channels"
    assert torch.equal(buf[..., :C].float().cpu(), x_t.permute(0, 2, 3, 1).to(tdt).float().cpu())          # kind 0: x_in = x_t
    call("mdm_zero_pad_channels", DT[dt], ptr(buf), N * H * W, C, Cp, stream())
    torch.cuda.synchronize()
    assert float(buf[..., C:].float().abs().sum()) == 0.0 and torch.equal(buf[..., :C].float().cpu(), x_t.permute(0, 2, 3, 1).to(tdt).float().cpu())
    with pytest.raises(RuntimeError, match="zero_pad_channels"):
        call("mdm_zero_pad_channels", DT[dt], ptr(buf), N * H * W, C, 7, stream())


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_softmax_colsum_pool_layout_temb(dt):
    from mdm import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    S = _q(torch.randn(96, 64, generator=g) * 3, dt).requires_grad_(True)
    P = torch.softmax(S, -1)
    gp = _q(torch.randn(96, 64, generator=g), dt)
    P.backward(gp)
    Sd = _up(S.detach(), dt)
    ops.softmax_fwd(DT[dt], Sd, 96, 64)
    torch.cuda.synchronize()
    assert _relerr(Sd, P.detach()) < _tol(dt, 0.5)
    gd = _up(gp, dt)
    ops.softmax_bwd(DT[dt], _up(P.detach(), dt), gd, 96, 64)
    torch.cuda.synchronize()
    assert _relerr(gd, S.grad) < _tol(dt)
    # column sums
    dY = _q(torch.randn(3, 50, 72, generator=g), dt)
    per = torch.ones(3, 80, device=dev)
    dbias = torch.ones(72, device=dev)
    ops.colsum(DT[dt], _up(dY, dt), 3, 50, 72, per_img=per[:, 8:], ld=80, acc_img=1, dbias=dbias)
    torch.cuda.synchronize()
    assert _relerr(per[:, 8:] - 1, dY.sum(1)) < 1e-5
    assert _relerr(dbias - 1, dY.sum((0, 1))) < 1e-5
    assert float(per[:, :8].sum()) == 24.0
    # 2x2 sum pool
    up = _q(torch.randn(2, 8, 8, 16, generator=g), dt)
    dst = torch.empty(2, 4, 4, 16, device=dev, dtype=_up(up, dt).dtype)
    ops.sumpool2(DT[dt], _up(up, dt), dst, 0, 2, 4, 4, 16)
    torch.cuda.synchronize()
    want = up.reshape(2, 4, 2, 4, 2, 16).sum((2, 4))
    assert _relerr(dst, want) < _tol(dt, 0.5)
    # layout converters round-trip with channel padding
    x = _q(torch.randn(2, 3, 8, 8, generator=g), dt)
    xh = torch.empty(2, 8, 8, 8, device=dev, dtype=dst.dtype)
    ops.nchw_to_nhwc(DT[dt], x.to(dev), xh, 2, 3, 8, 8, 8)
    back = torch.empty(2, 3, 8, 8, device=dev)
    ops.nhwc_to_nchw(DT[dt], xh, back, 2, 3, 8, 8, 8)
    torch.cuda.synchronize()
    assert torch.equal(back.cpu(), x) and float(xh[..., 3:].float().abs().sum()) == 0.0
    if dt == "f32":
        from oracle.unet_ref import timestep_embedding
        t = torch.tensor([1.0, 17.0, 500.0, 1000.0])
        y = torch.empty(4, 128, device=dev)
        ops.timestep_embedding(t.to(dev), 4, 128, y)
        torch.cuda.synchronize()
        assert (y.cpu() - timestep_embedding(t, 128)).abs().max() < 2e-4   # sin/cos of arguments up to 1e3


def test_grouped_weight_gradients_match_self_contained_ones():
    """mdm_wgrad_group_* (mdm_hip.h): many weight gradients of mixed shapes (big / small tiles, split and unsplit,
    concat, stride 2, folded upsample, more than one table of 96 split-K segments) as ONE grouped launch give the same
    gradients and bias gradients as one mdm_gemm call each; overwrite and accumulate forms."""
    from mdm import _lib, ops
    dt = "bf16"
    g = torch.Generator().manual_seed(77)
    shapes = [(8, 8, 64, 0, 64, 1, 0), (8, 8, 64, 0, 128, 1, 0), (4, 16, 64, 64, 64, 1, 0), (4, 32, 128, 0, 128, 1, 0),
              (8, 16, 64, 0, 64, 2, 0), (4, 8, 64, 0, 64, 1, 1), (16, 4, 256, 0, 256, 1, 0),      # N, H, C0, C1, Cout, stride, ups
              # all nine taps in one pass (wgrad_taps_body: 128 | Cout, 64 | Cin, maps 8 / 16 / 32 wide): concat, folded upsample,
              # two output-channel tiles, tiles cut over several CUs (partial slots + tile_parts_reduce_kernel)
              (4, 8, 64, 64, 128, 1, 0), (4, 16, 64, 64, 128, 1, 0), (2, 16, 128, 0, 128, 1, 1), (4, 4, 64, 0, 128, 1, 1),
              (2, 32, 64, 0, 256, 1, 0),
              (1, 64, 64, 64, 128, 1, 0), (2, 64, 128, 0, 128, 1, 0),    # 64-wide maps (cfg3): a 64-pixel slab is ONE image row; concat
              (4, 16, 8, 0, 128, 1, 0), (4, 16, 128, 0, 8, 1, 0)]          # the 8-channel ends of the net: one partly filled tile
    jobs = []
    for rep in range(240):
        N, H, C0, C1, Cout, stride, ups = shapes[rep % len(shapes)]
        pads = (1, 1, 1, 1) if stride == 1 else (0, 0, 1, 1)
        geom = ops.ConvGeom(N=N, IH=H, IW=H, C0=C0, C1=C1, Cout=Cout, stride=stride, pad_t=pads[0], pad_l=pads[1], pad_b=pads[2],
                            pad_r=pads[3], ups=ups)
        x0 = _up(_nhwc(_q(torch.randn(N, C0, H, H, generator=g), dt)), dt)
        x1 = _up(_nhwc(_q(torch.randn(N, C1, H, H, generator=g), dt)), dt) if C1 else None
        gy = _up(_nhwc(_q(torch.randn(N, Cout, geom.OH, geom.OW, generator=g), dt)), dt)
        jobs.append((geom, x0, x1, gy, rep % 5 == 0))
    want = []
    for geom, x0, x1, gy, acc in jobs:
        nb = ops.conv_wgrad_ws_bytes(1, geom)
        ws = torch.empty(max(nb // 4, 16), device=_dev())
        gw = torch.full((9, geom.Cout, geom.Cin), 0.5, device=_dev())
        gb = torch.zeros(geom.Cout, device=_dev())
        ops.conv_wgrad(1, geom, gy, x0, x1, gw, ws=ws, dbias=gb, acc=1)              # self-contained: own reduce launch
        want.append((gw, gb))
    fields, got, keep = [], [], []
    n_split = 0
    for geom, x0, x1, gy, acc in jobs:
        gw = torch.full((9, geom.Cout, geom.Cin), 0.5, device=_dev())
        gb = torch.zeros(geom.Cout, device=_dev())
        wf = ops.wgrad_fields(1, geom, gy, x0, x1, gw, dbias=gb, acc=int(acc))
        assert _lib.wgrad_group_accepts(**wf)
        sk = max(ops.wgrad_group_split(geom, slabs_per_item=4), 1)                     # short items: most layers split
        wf["splitk"] = sk
        if sk > 1:
            n_split += 1
            ws = torch.full((sk * 9 * geom.Cout * geom.Cin,), float("nan"), device=_dev())
            wf["ws"], wf["ws_bytes"] = ws, ws.numel() * 4
            keep.append(ws)
        fields.append(wf); got.append((gw, gb, acc))
    assert n_split > 96
    old = os.environ.get("MDM_TAPS_MIN_SHARE")
    os.environ["MDM_TAPS_MIN_SHARE"] = "0"            # the nine-tap path is taken only where a CU's share is long: force it for this small group
    try:
        grp = _lib.WgradGroup(fields, _dev())
    finally:
        if old is None:
            del os.environ["MDM_TAPS_MIN_SHARE"]
        else:
            os.environ["MDM_TAPS_MIN_SHARE"] = old
    grp.launch()
    torch.cuda.synchronize()
    for (a, ab, acc), (b, bb) in zip(got, want):
        ref = b if acc else b - 0.5                                                   # overwrite form: no initial value
        assert float((a - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 2e-6
        assert float((ab - bb).abs().max()) <= 1e-5 * float(bb.abs().max()) + 1e-5
        assert float((b - 0.5).abs().max()) > 0.1                                     # non-trivial gradients


@pytest.mark.parametrize("case", [(8, 8, 256, 256, True, False), (16, 4, 128, 256, True, True), (4, 8, 128, 64, False, False),
                                  # conv_small_body + its register epilogue (128-channel superslabs, C / G = 8): with the column sums, without SiLU
                                  (32, 4, 256, 256, True, True), (8, 8, 256, 512, True, True), (12, 4, 256, 256, False, False),
                                  (8, 8, 512, 256, True, True), (4, 4, 2048, 64, True, False),       # 64-channel groups -> 64-wide tiles
                                  (8, 8, 256, 768, False, False, 1), (16, 4, 256, 768, False, False, 1)])   # 1x1 convolutions (conv_lin2 tiles)
def test_groupnorm_backward_fused_into_the_data_gradient(case):
    """4x4 / 8x8 maps: z = silu?(GroupNorm(x)) feeds a 3x3 conv; the conv's data gradient runs the GroupNorm backward in
    its epilogue (gnb_*): dx, dgamma, dbeta and the optional column sums against torch autograd."""
    from mdm import ops
    dt = "bf16"
    N, H, C, Cout, silu, with_sums = case[:6]
    ks = case[6] if len(case) > 6 else 3
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + C)
    x = _q(torch.randn(N, C, H, H, generator=g) * 1.2 + 0.3, dt).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    w = _q(torch.randn(Cout, C, ks, ks, generator=g) / (ks * C ** 0.5), dt)
    z = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    if silu:
        z = F.silu(z)
    y = F.conv2d(z, w, None, padding=ks // 2)
    gy = _q(torch.randn(y.shape, generator=g), dt)
    y.backward(gy)
    dev = _dev()
    pd = ks // 2
    geom = ops.ConvGeom(N=N, IH=H, IW=H, C0=C, C1=0, Cout=Cout, KH=ks, KW=ks, pad_t=pd, pad_l=pd, pad_b=pd, pad_r=pd)
    assert ops.conv_dgrad_t_can_fuse_gn_bwd(1, geom)
    xh = _up(_nhwc(x.detach()), dt)
    zb = torch.empty_like(xh); stats = torch.empty(N, 32, 2, device=dev)
    ws = torch.empty(N * (64 * 32 + 4 * C), device=dev)
    gd, bd = gamma.detach().to(dev), beta.detach().to(dev)
    ops.groupnorm_fwd(1, xh.view(N, H * H, C), C, None, 0, N, H * H, gd, bd, silu, zb.view(N, H * H, C), stats, ws)
    wT = _up(_w_tap(w).transpose(1, 2).contiguous(), dt)
    base = _q(torch.randn(N, C, H, H, generator=g), dt)
    dx = _up(_nhwc(base), dt).clone()                      # dx is ACCUMULATED onto an existing gradient when no sums are asked
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    gnb = dict(x=xh, stats=stats, gamma=gd, beta=bd, dgamma=dg, dbeta=db, G=32, silu=silu)
    acc = 1
    if with_sums:
        per = torch.full((N, C + 8), 3.0, device=dev); tot = torch.zeros(C, device=dev)
        gnb.update(sum_img=per[:, 8:], sum_ld=C + 8, sum_all=tot)
        acc = 0
    ops.conv_dgrad_t(1, geom, _up(_nhwc(gy), dt), wT, dx, acc, gnb=gnb)
    torch.cuda.synchronize()
    want = _nhwc(x.grad + (base if acc else 0))
    assert _relerr(dx, want) < _tol(dt, 1.5)
    assert _relerr(dg, gamma.grad) < _tol(dt) and _relerr(db, beta.grad) < _tol(dt)
    if with_sums:
        gx = _nhwc(x.grad)
        scale = float(gx.abs().sum((1, 2)).mean())
        assert float((per[:, 8:].cpu() - gx.sum((1, 2))).abs().max()) < _tol(dt, 0.5) * scale and float((per[:, :8] - 3.0).abs().sum()) == 0
        assert float((tot.cpu() - gx.sum((0, 1, 2))).abs().max()) < _tol(dt, 0.5) * scale * N


@pytest.mark.parametrize("case", [(8, 8, 128, 256, True), (16, 4, 256, 128, True), (4, 8, 64, 512, False),
                                  (32, 4, 256, 256, True), (8, 8, 512, 256, False), (12, 4, 128, 256, True),        # conv_small_body + register epilogue
                                  (4, 4, 64, 2048, True),                                           # 64-channel groups -> 64-wide tiles
                                  (8, 8, 256, 256, False, 1), (16, 4, 256, 256, True, 1)])          # 1x1 convolutions (conv_lin2 tiles)
def test_groupnorm_forward_fused_into_the_producing_conv(case):
    """4x4 / 8x8 maps: the conv's epilogue (bias + time-embedding row + residual, bf16 store) also writes
    silu?(GroupNorm(y)) and the (mean, rstd) statistics (gnf_*)."""
    from mdm import ops
    dt = "bf16"
    N, H, C, Cout, silu = case[:5]
    ks = case[5] if len(case) > 5 else 3
    g = torch.Generator().manual_seed(N * 100 + H + Cout)
    x = _q(torch.randn(N, C, H, H, generator=g), dt)
    w = _q(torch.randn(Cout, C, ks, ks, generator=g) / (ks * C ** 0.5), dt)
    b = torch.randn(Cout, generator=g)
    rv = torch.randn(N, Cout, generator=g)
    res = _q(torch.randn(N, Cout, H, H, generator=g), dt)
    gamma = 1 + 0.2 * torch.randn(Cout, generator=g)
    beta = 0.1 * torch.randn(Cout, generator=g)
    y = _q(F.conv2d(x, w, b, padding=ks // 2) + rv[:, :, None, None] + res, dt)    # what a separate GroupNorm launch would read
    z = F.group_norm(y, 32, gamma, beta, eps=1e-6)
    if silu:
        z = F.silu(z)
    dev = _dev()
    pd = ks // 2
    geom = ops.ConvGeom(N=N, IH=H, IW=H, C0=C, C1=0, Cout=Cout, KH=ks, KW=ks, pad_t=pd, pad_l=pd, pad_b=pd, pad_r=pd)
    out = torch.empty(N, H, H, Cout, device=dev, dtype=torch.bfloat16)
    zo = torch.empty_like(out); stats = torch.full((N, 32, 2), float("nan"), device=dev)
    ops.conv_fwd(1, geom, _up(_nhwc(x), dt), None, _up(_w_tap(w), dt), b.to(dev), out, rowvec=rv.to(dev), rv_ld=Cout,
                 resid=_up(_nhwc(res), dt), gnf=dict(out=zo, gamma=gamma.to(dev), beta=beta.to(dev), stats=stats, G=32, silu=silu))
    torch.cuda.synchronize()
    assert _relerr(out, _nhwc(y)) < _tol(dt)
    assert _relerr(zo, _nhwc(z)) < _tol(dt)
    yg = y.reshape(N, 32, -1)
    assert _relerr(stats[..., 0], yg.mean(-1)) < 2e-2 and _relerr(stats[..., 1], 1.0 / torch.sqrt(yg.var(-1, unbiased=False) + 1e-6)) < 2e-2


@pytest.mark.parametrize("N,L,C", [(3, 16, 256), (2, 64, 256), (2, 256, 128), (1, 1024, 128), (2, 64, 32), (1, 256, 64), (2, 48, 64)])
def test_fused_attention_forward_and_backward(N, L, C):
    """csrc/attn.hip (unet6.py:316-324 and its autograd backward): o = softmax(q k^T / sqrt(C)) v on the NHWC qkv tensor,
    online softmax over 64-key tiles (L = 1024: 16 tiles; L = 16 / 48: partial tiles), dq / dk / dv from the saved
    log-sum-exp -- against fp64 torch on the same bf16-rounded inputs."""
    import math
    from mdm import ops
    g = torch.Generator().manual_seed(100 + L + C)
    dev = _dev()
    qkv = _q(torch.randn(N, L, 3 * C, generator=g) * 1.5, "bf16")
    qkv[0, 0, C:2 * C] *= 6.0                      # one key that dominates some rows: the running max jumps inside the walk
    qkv = _q(qkv, "bf16")
    do = _q(torch.randn(N, L, C, generator=g), "bf16")
    scale = 1.0 / math.sqrt(C)
    assert ops.attn_supported(1, L, C)
    x = qkv.double().requires_grad_(True)
    q, k, v = x[..., :C], x[..., C:2 * C], x[..., 2 * C:]
    s = torch.einsum("nlc,nmc->nlm", q, k) * scale
    want_o = torch.einsum("nlm,nmc->nlc", torch.softmax(s, -1), v)
    (want_o * do.double()).sum().backward()
    want_lse = torch.logsumexp(s, -1)
    d_qkv = _up(qkv, "bf16")
    o = torch.full((N, L, C), float("nan"), device=dev, dtype=torch.bfloat16)
    lse = torch.full((N, L), float("nan"), device=dev)
    ops.attn_fwd(1, d_qkv, o, lse, N, L, C, scale)
    torch.cuda.synchronize()
    assert _relerr(o, want_o.detach().float()) < 1e-2, _relerr(o, want_o.detach().float())
    assert float((lse.cpu() - want_lse.detach().float()).abs().max()) < 2e-2
    dq = torch.full((N, L, 3 * C), float("nan"), device=dev, dtype=torch.bfloat16)
    delta = torch.full((N, L), float("nan"), device=dev)
    ops.attn_bwd(1, d_qkv, o, _up(do, "bf16"), lse, delta, dq, N, L, C, scale)
    torch.cuda.synchronize()
    gx = x.grad.float()
    for name, sl in (("dq", slice(0, C)), ("dk", slice(C, 2 * C)), ("dv", slice(2 * C, 3 * C))):
        err = _relerr(dq[..., sl], gx[..., sl])
        assert err < 2.5e-2, (name, err)
    assert bool(torch.isfinite(dq.float()).all())

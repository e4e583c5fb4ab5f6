"""Host-side launchers: build `mdm_gemm_desc`s and call the C ABI.

Every function enqueues kernels on torch's current stream and returns at once.
Tensors are NHWC (`[N, H, W, C]`, C a multiple of 8) unless stated otherwise.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import _lib
from ._lib import BF16, F32, call, ptr, stream


@dataclass
class ConvGeom:
    """Geometry of one convolution site (reference unet6.py:232-235, 257-272, 472-475)."""
    N: int
    IH: int          # physical source extent
    IW: int
    C0: int          # channels of source 0 / source 1 (concat, unet6.py:501)
    C1: int
    Cout: int
    KH: int = 3
    KW: int = 3
    stride: int = 1
    pad_t: int = 1
    pad_l: int = 1
    pad_b: int = 1
    pad_r: int = 1
    ups: int = 0     # read the source through a nearest x2 upsample

    @property
    def Cin(self):
        return self.C0 + self.C1

    @property
    def VH(self):    # logical (virtual) input extent
        return self.IH << self.ups

    @property
    def VW(self):
        return self.IW << self.ups

    @property
    def OH(self):
        return (self.VH + self.pad_t + self.pad_b - self.KH) // self.stride + 1

    @property
    def OW(self):
        return (self.VW + self.pad_l + self.pad_r - self.KW) // self.stride + 1

    @property
    def taps(self):
        return self.KH * self.KW


def conv_flops(g: ConvGeom):
    """Algorithmic FLOPs of this convolution (the same for forward, data- and weight-gradient)."""
    return 2.0 * g.N * g.OH * g.OW * g.Cout * g.Cin * g.taps


def conv_fwd(dt, g: ConvGeom, src0, src1, w, bias, out, rowvec=None, rv_ld=0, resid=None, out_f32=0, ws=None, gnf=None, w_split=None,
             f32_split=0):
    """out[N,OH,OW,Cout] = conv(concat(src0,src1), w[tap][Cout][Cin]) + bias + rowvec[n] + resid.
    Returns the descriptor.  `gnf` = dict(out, gamma, beta, stats, G, silu, eps): also the GroupNorm of the result in the
    same launch (only where conv_fwd_can_fuse_gn says so); `fuse_gn_fwd(desc, ...)` sets it on a RECORDED call afterwards."""
    return _lib.gemm(**conv_fwd_fields(dt, g, src0, src1, w, bias, out, rowvec, rv_ld, resid, out_f32, ws, gnf, w_split, f32_split))


def conv_fwd_pair(fields_a, fields_b):
    """Two independent forward convolutions (conv_fwd_fields each) as one call: -> (desc_a, desc_b)."""
    return _lib.gemm_pair(fields_a, fields_b)


def conv_fwd_fields(dt, g: ConvGeom, src0, src1, w, bias, out, rowvec=None, rv_ld=0, resid=None, out_f32=0, ws=None, gnf=None,
                    w_split=None, f32_split=0):
    """`w_split` (fp32 only): the same filters from the store's split shadow (mdm_split_shadow) -- where the convolution qualifies,
    its products then run on the bf16 matrix pipe as hi / lo pairs (mdm_gemm_desc.B_split); elsewhere it is ignored.  `f32_split`:
    the same permission for the convolutions that run on the register-staged fp32 kernel (1x1, stride 2, the two ends of the net)."""
    extra = {} if gnf is None else _gnf_fields(gnf)
    if w_split is not None:
        extra["B_split"] = w_split
    if f32_split:
        extra["f32_split"] = 1
    return dict(**extra, dtype=dt, layout=0, M=g.N * g.OH * g.OW, N=g.Cout, K=g.taps * g.Cin,
              conv=1, OH=g.OH, OW=g.OW, IH=g.VH, IW=g.VW, KH=g.KH, KW=g.KW, stride=g.stride,
              pad_t=g.pad_t, pad_l=g.pad_l, transposed=0, ups=g.ups, C0=g.C0, C1=g.C1, Ck=g.Cin,
              src0=src0, src1=src1, ld0=g.C0, ld1=g.C1, B=w, ldb=g.Cin, wtap=g.Cout * g.Cin,
              D0=out, ldd0=g.Cout, N0=g.Cout, out_f32=out_f32, bias=bias, rowvec=rowvec, rv_ld=rv_ld,
              rows_per_img=g.OH * g.OW, resid=resid, ldr=g.Cout, _flops=conv_flops(g),
              ws=ws, ws_bytes=(ws.numel() * 4 if ws is not None else 0))


def _gnf_fields(gnf):
    return dict(gnf_out=gnf["out"], gnf_gamma=gnf["gamma"], gnf_beta=gnf["beta"], gnf_stats=gnf["stats"], gnf_G=int(gnf.get("G", 32)),
                gnf_silu=int(bool(gnf["silu"])), gnf_eps=float(gnf.get("eps", 1e-6)))


def conv_fwd_can_fuse_gn(desc, G=32):
    """True if the forward conv described by `desc` (as returned by conv_fwd) may carry the gnf_* epilogue."""
    return bool(_lib.load().mdm_gemm_can_fuse_gn_fwd(_lib.C.byref(desc), int(G)))


def fuse_gn_fwd(desc, gnf):
    """Attach the GroupNorm-forward epilogue to an already RECORDED conv_fwd call (its descriptor is replayed by reference)."""
    for k, v in _gnf_fields(gnf).items():
        setattr(desc, k, v.data_ptr() if hasattr(v, "data_ptr") else v)


def conv_dgrad(dt, g: ConvGeom, dy, w, dst0, acc0, dst1=None, acc1=0):
    """Gradient w.r.t. the (virtual) input: dst[N,VH,VW,C0|C1] (=|+=) conv_transpose(dy, w)."""
    _lib.gemm(dtype=dt, layout=1, M=g.N * g.VH * g.VW, N=g.Cin, K=g.taps * g.Cout,
              conv=1, OH=g.VH, OW=g.VW, IH=g.OH, IW=g.OW, KH=g.KH, KW=g.KW, stride=g.stride,
              pad_t=g.pad_t, pad_l=g.pad_l, transposed=1, ups=0, C0=g.Cout, C1=0, Ck=g.Cout,
              src0=dy, src1=None, ld0=g.Cout, ld1=0, B=w, ldb=g.Cin, wtap=g.Cout * g.Cin,
              D0=dst0, ldd0=g.C0, D1=dst1, ldd1=g.C1, N0=g.C0, acc0=acc0, acc1=acc1, _flops=conv_flops(g))


def _dgrad_t_fields(dt, g, dy, wT, dst0, acc0, dst1, acc1, ws):
    return dict(dtype=dt, layout=0, M=g.N * g.VH * g.VW, N=g.Cin, K=g.taps * g.Cout,
                conv=1, OH=g.VH, OW=g.VW, IH=g.OH, IW=g.OW, KH=g.KH, KW=g.KW, stride=g.stride,
                pad_t=g.pad_t, pad_l=g.pad_l, transposed=1, ups=0, C0=g.Cout, C1=0, Ck=g.Cout,
                src0=dy, src1=None, ld0=g.Cout, ld1=0, B=wT, ldb=g.Cout, wtap=g.Cout * g.Cin,
                D0=dst0, ldd0=g.C0, D1=dst1, ldd1=g.C1, N0=g.C0, acc0=acc0, acc1=acc1, _flops=conv_flops(g),
                ws=ws, ws_bytes=(ws.numel() * 4 if ws is not None else 0))


def conv_dgrad_t(dt, g: ConvGeom, dy, wT, dst0, acc0, dst1=None, acc1=0, ws=None, gnb=None):
    _lib.gemm(**conv_dgrad_t_fields(dt, g, dy, wT, dst0, acc0, dst1, acc1, ws, gnb))


def conv_dgrad_t_fields(dt, g: ConvGeom, dy, wT, dst0, acc0, dst1=None, acc1=0, ws=None, gnb=None):
    """Same gradient as conv_dgrad but with per-tap TRANSPOSED filters wT[tap][Cin][Cout]: both operands
    are k-contiguous (layout 0), the path the forward uses.  `gnb` (see conv_dgrad_t_can_fuse_gn_bwd): the conv's input
    was z = silu?(GroupNorm(x)); dict(x, stats, gamma, beta, dgamma, dbeta, G, silu[, sum_img, sum_ld, sum_all]) makes
    the epilogue run that GroupNorm's backward, so dst0 receives dx (not dz) and dgamma / dbeta are accumulated."""
    f = _dgrad_t_fields(dt, g, dy, wT, dst0, acc0, dst1, acc1, ws)
    if gnb is not None:
        f.update(gnb_x=gnb["x"], gnb_stats=gnb["stats"], gnb_gamma=gnb["gamma"], gnb_beta=gnb["beta"],
                 gnb_dgamma=gnb["dgamma"], gnb_dbeta=gnb["dbeta"], gnb_G=int(gnb.get("G", 32)), gnb_silu=int(bool(gnb["silu"])),
                 gnb_sum_img=gnb.get("sum_img"), gnb_sum_ld=int(gnb.get("sum_ld", 0)), gnb_sum_all=gnb.get("sum_all"),
                 gnb_add=gnb.get("add"))
    return f


def conv_dgrad_t_can_fuse_gn_bwd(dt, g: ConvGeom, G=32):
    """True if conv_dgrad_t on this geometry runs on the whole-image halo tiles, so that `gnb` may be used."""
    if g.C1 or g.ups:
        return False
    f = _dgrad_t_fields(dt, g, 16, 16, 16, 0, None, 0, None)
    f.pop("_flops")
    return bool(_lib.load().mdm_gemm_can_fuse_gn_bwd(_lib.C.byref(_lib._desc(f)), int(G)))


def wgrad_fields(dt, g, dy, src0, src1, dw, splitk=0, ws=None, dbias=None, acc=1):
    """Descriptor fields of a weight gradient (for `conv_wgrad`, or for a `_lib.WgradGroup`)."""
    return dict(dtype=dt, layout=2, M=g.Cout, N=g.Cin, K=g.N * g.OH * g.OW,
                conv=1, OH=g.OH, OW=g.OW, IH=g.VH, IW=g.VW, KH=g.KH, KW=g.KW, stride=g.stride,
                pad_t=g.pad_t, pad_l=g.pad_l, transposed=0, ups=g.ups, C0=g.C0, C1=g.C1, Ck=g.Cin,
                src0=src0, src1=src1, ld0=g.C0, ld1=g.C1, A=dy, lda=g.Cout,
                D0=dw, ldd0=g.Cin, N0=g.Cin, out_f32=1, acc0=int(acc), splitk=splitk, dtap=g.Cout * g.Cin, _flops=conv_flops(g),
                ws=ws, ws_bytes=(ws.numel() * 4 if ws is not None else 0), dbias=dbias)


def conv_wgrad(dt, g: ConvGeom, dy, src0, src1, dw, splitk=0, ws=None, dbias=None, acc=1):
    """dw[tap][Cout][Cin] (fp32) (+)= sum_pixels dy x gathered input.  `ws`: fp32 split-K workspace tensor;
    `dbias` (bf16 path only): fp32 [Cout] that also receives += column sums of dy."""
    _lib.gemm(**wgrad_fields(dt, g, dy, src0, src1, dw, splitk, ws, dbias, acc))


def conv_wgrad_ws_bytes(dt, g: ConvGeom):
    """Workspace bytes conv_wgrad wants for this geometry under mdm_gemm's own split rule (0: no partial slabs)."""
    f = wgrad_fields(dt, g, 16, 16, 16 if g.C1 else None, 16)    # dummy non-null pointers: no launch
    return _lib.gemm_plan(**f)[1]


def wgrad_group_split(g: ConvGeom, slabs_per_item=48):
    """k-splits of a weight gradient that runs inside a group: the group fills the chip, so a layer is only cut
    into work items of ~slabs_per_item 64-pixel slabs (never below 8) -- not into as many as fill 256 CUs alone."""
    nslabs = (g.N * g.OH * g.OW) // 64
    return max(1, min(int(round(nslabs / float(slabs_per_item))), nslabs // 8))


def matmul(dt, layout, M, N, K, A, lda, B, ldb, D, ldd, batch=1, sA=0, sB=0, sD=0, alpha=1.0, bias=None,
           acc=0, out_f32=0, splitk=1, dbias=None, ws=None):
    """Plain (batched) contraction in one of the three layouts (see mdm_hip.h).  dbias (layout 2): += column sums of A.
    splitk > 1 needs `ws` (fp32 tensor, >= splitk * M * N elements): without a workspace the reduction is not split."""
    _lib.gemm(dtype=dt, layout=layout, M=M, N=N, K=K, batch=batch, sA=sA, sB=sB, sD=sD, A=A, lda=lda, B=B, ldb=ldb,
              D0=D, ldd0=ldd, N0=N, alpha=alpha, bias=bias, acc0=acc, out_f32=out_f32, splitk=splitk, dbias=dbias,
              ws=ws, ws_bytes=(ws.numel() * 4 if ws is not None else 0))


def skinny_supported(M, N, K, splits=1):
    return bool(_lib.load().mdm_skinny_supported(M, N, K, splits))


def skinny_linear_fwd(x, W, bias, M, N, K, y, act_out=None, t=None, variant=None, emb_out=None):
    """y = x W^T + bias (fp32, M = batch rows), act_out = silu(y); x=None: the input is the timestep embedding of t."""
    flip, shift = (0, 1.0) if variant is None else (int(bool(variant[0])), float(variant[1]))
    call("mdm_skinny_linear_fwd", ptr(x), K, ptr(t), flip, shift, ptr(emb_out), ptr(W), K, ptr(bias), M, N, K, ptr(y), N,
         ptr(act_out), stream())


def skinny_linear_bwd(dy, W, M, N, K, dx=None, pre=None, splits=1, slabs=None):
    """dx = dy W (* silu'(pre)); splits > 1: partial sums to slabs[splits][M][N] for silu_bwd_sum."""
    call("mdm_skinny_linear_bwd", ptr(dy), K, ptr(W), N, M, N, K, splits, ptr(pre), ptr(dx), ptr(slabs), stream())


def silu_bwd_sum(pre, slabs, nslab, n, dx):
    call("mdm_silu_bwd_sum", ptr(pre), ptr(slabs), nslab, n, ptr(dx), stream())


def groupnorm_fwd(dt, src0, C0, src1, C1, N, P, gamma, beta, silu, y, stats, ws, G=32, eps=1e-6):
    call("mdm_groupnorm_fwd", dt, ptr(src0), C0, ptr(src1), C1, N, P, G, eps, ptr(gamma), ptr(beta), int(silu),
         ptr(y), ptr(stats), ptr(ws), stream())


def groupnorm_bwd(dt, src0, C0, src1, C1, N, P, gamma, beta, silu, dy, stats, dst0, acc0, dst1, acc1,
                  dgamma, dbeta, ws, G=32, sum_img=None, sum_ld=0, sum_all=None, add0=None, add1=None):
    """dst = (acc ? dst : add or 0) + dx.  add0 / add1: tensors laid out like dst0 / dst1 that are added without being
    modified (the gradient arriving over a residual branch)."""
    need = _lib.load().mdm_groupnorm_bwd_ws_floats(dt, N, C0 + C1)
    if need and (ws is None or ws.numel() < need):
        raise ValueError(f"groupnorm_bwd: workspace of {0 if ws is None else ws.numel()} floats, {need} needed")
    a0 = dst0 if acc0 else add0
    a0b = add0 if acc0 else None            # accumulate AND a pending addend: both are added
    assert not (acc1 and add1 is not None), "groupnorm_bwd: the second source takes one addend"
    a1 = dst1 if acc1 else add1
    call("mdm_groupnorm_bwd_add", dt, ptr(src0), C0, ptr(src1), C1, N, P, G, ptr(gamma), ptr(beta), int(silu), ptr(dy),
         ptr(stats), ptr(dst0), ptr(a0), ptr(dst1), ptr(a1), ptr(dgamma), ptr(dbeta), ptr(sum_img), sum_ld,
         ptr(sum_all), ptr(ws), ptr(a0b), stream())


def attn_supported(dt, L, C):
    return bool(_lib.load().mdm_attn_supported(dt, L, C))


def attn_fwd(dt, qkv, o, lse, N, L, C, scale):
    call("mdm_attn_fwd", dt, ptr(qkv), ptr(o), ptr(lse), N, L, C, float(scale), stream())


def attn_f32_small_supported(L, C):
    return bool(_lib.load().mdm_attn_f32_small_supported(L, C))


def attn_f32_small_fwd(qkv, o, S, N, L, C, scale):
    """Exact-fp32 fused attention forward for L <= 64; S receives the probabilities (the unfused backward reads them)."""
    call("mdm_attn_f32_small_fwd", ptr(qkv), ptr(o), ptr(S), N, L, C, float(scale), stream())


def attn_bwd(dt, qkv, o, do, lse, delta, dqkv, N, L, C, scale):
    call("mdm_attn_bwd", dt, ptr(qkv), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dqkv), N, L, C, float(scale), stream())


def softmax_fwd(dt, S, rows, L):
    call("mdm_softmax_fwd", dt, ptr(S), rows, L, stream())


def softmax_bwd(dt, P, dP, rows, L):
    call("mdm_softmax_bwd", dt, ptr(P), ptr(dP), rows, L, stream())


def timestep_embedding(t, N, dim, y, variant=None):
    if variant is None:
        call("mdm_timestep_embedding", ptr(t), N, dim, ptr(y), stream())
    else:
        call("mdm_timestep_embedding2", ptr(t), N, dim, int(bool(variant[0])), float(variant[1]), ptr(y), stream())


def attn_mh_fwd(dt, q, k, v, o, lse, N, L, C, heads, scale):
    call("mdm_attn_mh_fwd", dt, ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), N, L, C, heads, float(scale), stream())


def attn_mh_bwd(dt, q, k, v, o, do, lse, delta, dq, dk, dv, N, L, C, heads, scale):
    call("mdm_attn_mh_bwd", dt, ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv),
         N, L, C, heads, float(scale), stream())


def silu_fwd(x, y, n):
    call("mdm_silu_fwd", ptr(x), ptr(y), n, stream())


def silu_bwd(x, dy, dx, acc, n):
    call("mdm_silu_bwd", ptr(x), ptr(dy), ptr(dx), int(acc), n, stream())


def colsum(dt, dY, N, P, C, per_img=None, ld=0, acc_img=0, dbias=None):
    call("mdm_colsum", dt, ptr(dY), N, P, C, ptr(per_img), ld, int(acc_img), ptr(dbias), stream())


def sumpool2(dt, g, dst, acc, N, H, W, C):
    call("mdm_sumpool2", dt, ptr(g), ptr(dst), int(acc), N, H, W, C, stream())


def nchw_to_nhwc(dt, x, y, N, C, H, W, Cp):
    call("mdm_nchw_to_nhwc", dt, ptr(x), ptr(y), N, C, H, W, Cp, stream())


def nhwc_to_nchw(dt, x, y, N, C, H, W, Cp):
    call("mdm_nhwc_to_nchw", dt, ptr(x), ptr(y), N, C, H, W, Cp, stream())


def add_(dt, dst, src):
    call("mdm_add", dt, ptr(dst), ptr(src), dst.numel(), stream())


def add3(dt, dst, x, y=None):
    """dst = x + y (y None: a copy)."""
    call("mdm_add3", dt, ptr(dst), ptr(x), ptr(y), dst.numel(), stream())


def fill(t, v):
    call("mdm_fill_f32", ptr(t), float(v), t.numel(), stream())


def fill_segments(t, segs, v=0.0):
    """t[off:off+len] = v for the (off, len) rows of the device int64 table `segs`."""
    call("mdm_fill_segments_f32", ptr(t), ptr(segs), int(segs.shape[0]), float(v), stream())


def cast_bf16(src, dst):
    call("mdm_cast_bf16", ptr(src), ptr(dst), src.numel(), stream())

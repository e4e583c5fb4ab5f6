"""CPU oracle: the unet6 U-Net (TEST INFRASTRUCTURE ONLY).

Functional fp32 restatement of /root/reference/code/models/unet/unet6.py
(`UNet` :365-506 and its blocks) driven by a plain {name: tensor} dict that uses
the reference's state_dict key grammar, so weights
interchange with the reference and with the HIP model.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def unet6_config(image_size: int, in_channels: int = 3, out_channels: int = 3):
    """models_Unet.py:132-171 -- the `Model('unet6', C, H, W, out_C)` presets."""
    if image_size in (32, 64):
        mult, attn = [1, 2, 2, 2], [False, False, True, False]
    elif image_size in (128, 256):
        mult, attn = [1, 1, 2, 2, 4, 4], [False, False, False, False, True, False]
    else:
        raise NotImplementedError("model selection error")
    return dict(in_channels=in_channels, hid_channels=128, out_channels=out_channels,
                ch_multipliers=mult, num_res_blocks=2, apply_attn=attn)


def param_shapes(cfg):
    """Ordered {key: shape} in the reference's registration order (unet6.py:395-415)."""
    cin, hid, cout = cfg["in_channels"], cfg["hid_channels"], cfg["out_channels"]
    mult, nres, attn = cfg["ch_multipliers"], cfg["num_res_blocks"], cfg["apply_attn"]
    temb = cfg.get("time_embedding_dim") or 4 * hid
    levels = len(mult)
    out = {}

    def lin(p, i, o):
        out[p + ".weight"] = (o, i)
        out[p + ".bias"] = (o,)

    def conv(p, i, o, k):
        out[p + ".weight"] = (o, i, k, k)
        out[p + ".bias"] = (o,)

    def norm(p, c):
        out[p + ".weight"] = (c,)
        out[p + ".bias"] = (c,)

    def res(p, i, o):                       # unet6.py:340-354
        norm(p + ".norm1", i); conv(p + ".conv1", i, o, 3); lin(p + ".fc", temb, o)
        norm(p + ".norm2", o); conv(p + ".conv2", o, o, 3)
        if i != o:
            conv(p + ".skip", i, o, 1)

    def att(p, c):                          # unet6.py:299-314
        norm(p + ".norm", c); conv(p + ".project_in", c, 3 * c, 1); conv(p + ".project_out", c, c, 1)

    def block(p, i, o, a):                  # unet6.py:417-427
        if a:
            res(p + ".0", i, o); att(p + ".1", o)
        else:
            res(p, i, o)

    lin("embed.0", hid, temb); lin("embed.2", temb, temb)
    conv("in_conv", cin, hid, 3)
    for l in range(levels):                 # unet6.py:429-444
        prev = (mult[l - 1] if l else 1) * hid
        cur = mult[l] * hid
        block(f"downsamples.level_{l}.0", prev, cur, attn[l])
        for j in range(1, nres):
            block(f"downsamples.level_{l}.{j}", cur, cur, attn[l])
        if l != levels - 1:
            conv(f"downsamples.level_{l}.{nres}.1", cur, cur, 3)
    mid = mult[-1] * hid
    res("middle.0", mid, mid); att("middle.1", mid); res("middle.2", mid, mid)
    chs = [hid * m for m in mult]
    for l in range(levels):                 # unet6.py:446-476 (registered level_0 first)
        nxt = hid if l == 0 else chs[l - 1]
        prv = chs[-1] if l == levels - 1 else chs[l + 1]
        cur = chs[l]
        block(f"upsamples.level_{l}.0", prv + cur, cur, attn[l])
        for j in range(1, nres):
            block(f"upsamples.level_{l}.{j}", 2 * cur, cur, attn[l])
        block(f"upsamples.level_{l}.{nres}", nxt + cur, cur, attn[l])
        if l != 0:
            conv(f"upsamples.level_{l}.{nres + 1}.1", cur, cur, 3)
    norm("out_conv.0", hid); conv("out_conv.2", hid, cout, 3)
    return out


def random_params(cfg, seed=1234, scale=None):
    """Deterministic non-degenerate weights for every key (re-randomises the
    reference's `init_scale=0` layers).  Same recipe in tests and fixtures."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for k, shp in param_shapes(cfg).items():
        if len(shp) == 1:
            if k.endswith("weight"):        # norm gamma
                p[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
            else:
                p[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            s = scale if scale is not None else 1.0 / math.sqrt(fan_in)
            p[k] = s * torch.randn(shp, generator=g)
    return p


def timestep_embedding(t, dim):
    """unet6.py:18-34."""
    half = dim // 2
    freq = torch.exp(-torch.arange(half, dtype=torch.float32) * (math.log(10000) / (half - 1)))
    ang = torch.outer(t.ravel().to(torch.float32), freq)
    e = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if dim % 2 == 1:
        e = F.pad(e, [0, 1])
    return e


def _gn(x, p, pre):
    return F.group_norm(x, 32, p[pre + ".weight"], p[pre + ".bias"], eps=1e-6)   # unet6.py:291-293


def _conv(x, p, pre, stride=1, padding=0):
    return F.conv2d(x, p[pre + ".weight"], p[pre + ".bias"], stride=stride, padding=padding)


def res_block(x, temb, p, pre):
    """unet6.py:356-362 (dropout p=0)."""
    skip = _conv(x, p, pre + ".skip") if (pre + ".skip.weight") in p else x
    h = _conv(F.silu(_gn(x, p, pre + ".norm1")), p, pre + ".conv1", padding=1)
    h = h + F.linear(F.silu(temb), p[pre + ".fc.weight"], p[pre + ".fc.bias"])[:, :, None, None]
    h = _conv(F.silu(_gn(h, p, pre + ".norm2")), p, pre + ".conv2", padding=1)
    return h + skip


def attn_block(x, p, pre):
    """unet6.py:316-333: one head over L=H*W tokens, scale 1/sqrt(C)."""
    b, c, hh, ww = x.shape
    q, k, v = _conv(_gn(x, p, pre + ".norm"), p, pre + ".project_in").chunk(3, dim=1)
    q = q.reshape(b, c, hh * ww); k = k.reshape(b, c, hh * ww); v = v.reshape(b, c, hh * ww)
    w = torch.softmax(torch.einsum("bcl,bcm->blm", q, k) / math.sqrt(c), dim=-1)
    o = torch.einsum("blm,bcm->bcl", w, v).reshape(b, c, hh, ww)
    return _conv(o, p, pre + ".project_out") + x


def same_pad_stride2(x):
    """SamePad2d(3,2) (unet6.py:257-272): for even H,W pads 0 top/left, 1 bottom/right."""
    _, _, h, w = x.shape
    hp = 2 * math.ceil(h / 2 - 1) + 3 - h
    wp = 2 * math.ceil(w / 2 - 1) + 3 - w
    return F.pad(x, (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2))


def unet_forward(p, cfg, x, t):
    """unet6.py:478-506."""
    hid, mult, nres, attn = cfg["hid_channels"], cfg["ch_multipliers"], cfg["num_res_blocks"], cfg["apply_attn"]
    levels = len(mult)
    temb = timestep_embedding(t, hid).to(p["embed.0.weight"].dtype)     # (fp32 upstream; fp64 only for the yardstick runs below)
    temb = F.linear(temb, p["embed.0.weight"], p["embed.0.bias"])
    temb = F.linear(F.silu(temb), p["embed.2.weight"], p["embed.2.bias"])

    def block(h, pre, a):
        if a:
            return attn_block(res_block(h, temb, p, pre + ".0"), p, pre + ".1")
        return res_block(h, temb, p, pre)

    hs = [_conv(x, p, "in_conv", padding=1)]
    for l in range(levels):
        for j in range(nres):
            hs.append(block(hs[-1], f"downsamples.level_{l}.{j}", attn[l]))
        if l != levels - 1:
            hs.append(_conv(same_pad_stride2(hs[-1]), p, f"downsamples.level_{l}.{nres}.1", stride=2))
    h = res_block(hs[-1], temb, p, "middle.0")
    h = attn_block(h, p, "middle.1")
    h = res_block(h, temb, p, "middle.2")
    for l in range(levels - 1, -1, -1):
        for j in range(nres + 1):
            h = block(torch.cat([h, hs.pop()], dim=1), f"upsamples.level_{l}.{j}", attn[l])
        if l != 0:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = _conv(h, p, f"upsamples.level_{l}.{nres + 1}.1", padding=1)
    h = F.silu(_gn(h, p, "out_conv.0"))
    return _conv(h, p, "out_conv.2", padding=1)


class UNetRef(torch.nn.Module):
    """nn.Module wrapper: `model(x, t).sample`, `.device`, `.parameters()`."""

    class _Out:
        def __init__(self, sample):
            self.sample = sample

    def __init__(self, cfg, params=None, seed=1234, dtype=torch.float32):
        """dtype=torch.float64: the same network in double precision -- the YARDSTICK the parity tests measure conditioning
        with (how far the reference's own fp32 arithmetic is from the exact result on a given fixture), never a target."""
        super().__init__()
        self.cfg = dict(cfg)
        self.dtype = dtype
        params = params if params is not None else random_params(cfg, seed)
        self.keys = list(params.keys())
        self.plist = torch.nn.ParameterList([torch.nn.Parameter(params[k].clone().to(dtype)) for k in self.keys])

    @property
    def device(self):
        return self.plist[0].device

    def pdict(self):
        return {k: v for k, v in zip(self.keys, self.plist)}

    def forward(self, x, t):
        return UNetRef._Out(unet_forward(self.pdict(), self.cfg, x.to(self.dtype), t))

"""CPU oracle: the `diffusers.UNet2DModel`-shaped U-Net (TEST INFRASTRUCTURE ONLY).

diffusers is absent offline: this is a functional fp32 restatement of UNet2DModel's PUBLISHED forward
(defaults as instantiated by reference code/utils/model.py:24-32) driven by a {diffusers key: tensor} dict --
**parity unpinned**: nothing here has been run against the package.  It exists so that the HIP assembly
(mdm/unet2d.py) is checked by an independent implementation written with plain torch ops and autograd.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def timesteps_embedding(t, dim):
    """`Timesteps(num_channels=dim, flip_sin_to_cos=True, downscale_freq_shift=0)`."""
    half = dim // 2
    freq = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    ang = t.reshape(-1, 1).float() * freq[None]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def _gn(x, p, pre, eps):
    return F.group_norm(x, 32, p[pre + ".weight"], p[pre + ".bias"], eps=eps)


def resnet(x, temb, p, pre, eps):
    """ResnetBlock2D (time_embedding_norm='default', output_scale_factor=1)."""
    h = F.conv2d(F.silu(_gn(x, p, pre + ".norm1", eps)), p[pre + ".conv1.weight"], p[pre + ".conv1.bias"], padding=1)
    h = h + F.linear(F.silu(temb), p[pre + ".time_emb_proj.weight"], p[pre + ".time_emb_proj.bias"])[:, :, None, None]
    h = F.conv2d(F.silu(_gn(h, p, pre + ".norm2", eps)), p[pre + ".conv2.weight"], p[pre + ".conv2.bias"], padding=1)
    if (pre + ".conv_shortcut.weight") in p:
        x = F.conv2d(x, p[pre + ".conv_shortcut.weight"], p[pre + ".conv_shortcut.bias"])
    return x + h


def attention(x, p, pre, head_dim, eps):
    """Attention with residual_connection, heads = C / head_dim, scale = head_dim ** -0.5 (rescale_output_factor 1)."""
    n, c, hh, ww = x.shape
    heads = c // head_dim
    h = _gn(x, p, pre + ".group_norm", eps).reshape(n, c, hh * ww).transpose(1, 2)       # [N, L, C]
    q = F.linear(h, p[pre + ".to_q.weight"], p[pre + ".to_q.bias"])
    k = F.linear(h, p[pre + ".to_k.weight"], p[pre + ".to_k.bias"])
    v = F.linear(h, p[pre + ".to_v.weight"], p[pre + ".to_v.bias"])
    split = lambda z: z.reshape(n, -1, heads, head_dim).transpose(1, 2)                  # [N, heads, L, d]
    w = torch.softmax(split(q) @ split(k).transpose(-1, -2) * head_dim ** -0.5, dim=-1)
    o = (w @ split(v)).transpose(1, 2).reshape(n, -1, c)
    o = F.linear(o, p[pre + ".to_out.0.weight"], p[pre + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(n, c, hh, ww) + x


def unet2d_forward(p, cfg, x, t):
    boc, lpb, hd, eps = list(cfg["block_out_channels"]), cfg["layers_per_block"], cfg["attention_head_dim"], cfg.get("norm_eps", 1e-5)
    temb = timesteps_embedding(t, boc[0])
    temb = F.linear(temb, p["time_embedding.linear_1.weight"], p["time_embedding.linear_1.bias"])
    temb = F.linear(F.silu(temb), p["time_embedding.linear_2.weight"], p["time_embedding.linear_2.bias"])
    h = F.conv2d(x, p["conv_in.weight"], p["conv_in.bias"], padding=1)
    hs = [h]
    for i in range(len(boc)):
        for j in range(lpb):
            h = resnet(h, temb, p, f"down_blocks.{i}.resnets.{j}", eps)
            if cfg["down_attn"][i]:
                h = attention(h, p, f"down_blocks.{i}.attentions.{j}", hd, eps)
            hs.append(h)
        if i != len(boc) - 1:
            h = F.conv2d(h, p[f"down_blocks.{i}.downsamplers.0.conv.weight"], p[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
            hs.append(h)
    h = resnet(h, temb, p, "mid_block.resnets.0", eps)
    h = attention(h, p, "mid_block.attentions.0", hd, eps)
    h = resnet(h, temb, p, "mid_block.resnets.1", eps)
    for i in range(len(boc)):
        for j in range(lpb + 1):
            h = resnet(torch.cat([h, hs.pop()], dim=1), temb, p, f"up_blocks.{i}.resnets.{j}", eps)
            if cfg["up_attn"][i]:
                h = attention(h, p, f"up_blocks.{i}.attentions.{j}", hd, eps)
        if i != len(boc) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, p[f"up_blocks.{i}.upsamplers.0.conv.weight"], p[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    h = F.silu(_gn(h, p, "conv_norm_out", eps))
    return F.conv2d(h, p["conv_out.weight"], p["conv_out.bias"], padding=1)

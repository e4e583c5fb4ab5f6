"""`diffusers.UNet2DModel`-shaped U-Net on libmdm_hip.so.

What upstream actually trains: reference code/utils/model.py:3-33 `MyModel(dim_channel, dim_height, dim_width,
num_attention)` = `UNet2DModel(sample_size, in_channels, out_channels, layers_per_block=2, block_out_channels=(128, 128,
256, 256, 512, 512), down_block_types, up_block_types)` with `num_attention` in 1..5 choosing which levels are
`AttnDownBlock2D` / `AttnUpBlock2D`; everything else is UNet2DModel's defaults (positional time embedding with
flip_sin_to_cos=True / freq_shift=0, GroupNorm(32, eps=1e-5), SiLU, attention_head_dim=8, downsample_padding=1,
add_attention=True in the mid block).

`diffusers` is not installed here and cannot be fetched: the architecture below follows the package's
PUBLISHED layout (module names = state_dict key grammar, forward order, defaults) and is **parity unpinned** -- it is
checked against this repo's own CPU restatement (oracle/unet2d_ref.py), not against diffusers.  Only the ASSEMBLY is
new: every block runs on the kernels of the unet6 path (conv / GroupNorm / time-embedding contraction / grouped weight
gradients); the attention blocks have C/8 heads of width 8, which is the multi-head VALU kernel `mdm_attn_mh_*`.

Differences from unet6 that the assembly has to express: GroupNorm eps 1e-5; [cos | sin] time embedding with exponent
/ half; stride-2 downsampling with symmetric padding 1 (unet6: SamePad2d, bottom/right only); attention = GroupNorm ->
three Linear projections (1x1 convolutions here) -> multi-head softmax(q k^T / sqrt(8)) v -> Linear -> + residual;
key grammar `down_blocks.i.resnets.j.*`, `...attentions.j.{group_norm,to_q,to_k,to_v,to_out.0}`, `...downsamplers.0.conv`,
`mid_block.*`, `up_blocks.i.*`, `conv_in`, `time_embedding.linear_{1,2}`, `conv_norm_out`, `conv_out`.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

from . import ops
from ._lib import BF16
from .unet import UNet, _Conv, _Norm, _pad8, _Temb


def my_model_config(dim_channel, dim_height, num_attention=1, block_out_channels=(128, 128, 256, 256, 512, 512),
                    layers_per_block=2, attention_head_dim=8):
    """reference utils/model.py:3-33 as a plain dict."""
    L = len(block_out_channels)
    placement = {1: [4], 2: [3, 4], 3: [2, 3, 4], 4: [1, 2, 3, 4], 5: [1, 2, 3, 4, 5]}
    if num_attention not in placement:
        raise NotImplementedError("not implemented")
    down = [i in placement[num_attention] for i in range(L)]
    # utils/model.py lists the up blocks explicitly: the attention levels mirrored (AttnUpBlock2D at up index L-1-i)
    up = [(L - 1 - i) in placement[num_attention] for i in range(L)]
    return dict(in_channels=dim_channel, out_channels=dim_channel, sample_size=dim_height, block_out_channels=tuple(block_out_channels),
                layers_per_block=layers_per_block, down_attn=down, up_attn=up, attention_head_dim=attention_head_dim, norm_eps=1e-5)


class _AttnMH:
    """softmax(q k^T / sqrt(d)) v with C/d heads on separate q, k, v tensors (diffusers Attention inside UNet2DModel)."""

    def __init__(self, net, q, k, v, out, heads):
        self.net, self.q, self.k, self.v, self.out, self.heads = net, q, k, v, out, heads

    def declare(self, st):
        pass

    def fwd(self):
        n, o = self.net, self.out
        N, L, C = o.N, o.P, o.C
        self.lse = n.alloc((N, self.heads, L), torch.float32)
        ops.attn_mh_fwd(n.dt, self.q.data, self.k.data, self.v.data, o.data, self.lse, N, L, C, self.heads,
                        1.0 / math.sqrt(C // self.heads))

    def bwd(self):
        n, o = self.net, self.out
        N, L, C = o.N, o.P, o.C
        do = n.grad_for_read(o)
        gs = []
        for a in (self.q, self.k, self.v):
            g, acc, _ = n.grad_for_write(a)
            assert acc == 0
            gs.append(g)
        self.delta = n.alloc((N, self.heads, L), torch.float32)
        ops.attn_mh_bwd(n.dt, self.q.data, self.k.data, self.v.data, o.data, do, self.lse, self.delta, gs[0], gs[1], gs[2],
                        N, L, C, self.heads, 1.0 / math.sqrt(C // self.heads))


class UNet2D(UNet):
    """`UNet2D(my_model_config(C, H, num_attention), N, H, W, dtype)`; same surface as `mdm.UNet`."""

    def _build_specs(self):
        cfg, N = self.cfg, self.N
        cin, cout, boc = cfg["in_channels"], cfg["out_channels"], list(cfg["block_out_channels"])
        lpb, hd, eps = cfg["layers_per_block"], cfg["attention_head_dim"], cfg.get("norm_eps", 1e-5)
        temb = 4 * boc[0]
        self.cin, self.cout, self.cin_p, self.cout_p = cin, cout, _pad8(cin), _pad8(cout)
        self.acts, self.specs, self.fc_slots = [], [], OrderedDict()
        self.ref_order = []
        G = ops.ConvGeom

        def conv(name, src0, src1, Cout, k=3, stride=1, ups=0, fc=None, resid=None, rshape=None):
            pads = (1, 1, 1, 1) if k == 3 and stride == 1 else (1, 1, 0, 0) if k == 3 else (0, 0, 0, 0)   # Downsample2D: padding=1
            g = G(N=N, IH=src0.H, IW=src0.W, C0=src0.C, C1=src1.C if src1 else 0, Cout=Cout, KH=k, KW=k, stride=stride,
                  pad_t=pads[0], pad_l=pads[1], pad_b=pads[2], pad_r=pads[3], ups=ups)
            out = self._act(name, g.OH, g.OW, Cout)
            c = _Conv(self, name, g, src0, src1, out, fc_slot=fc, resid=resid)
            c.rshape = rshape or (Cout, g.Cin, k, k)
            self.specs.append(c)
            return out

        def norm(name, src0, src1, silu):
            out = self._act(name, src0.H, src0.W, src0.C + (src1.C if src1 else 0))
            self.specs.append(_Norm(self, name, src0, src1, out, silu, eps=eps))
            out.norm_spec = self.specs[-1]
            return out

        def resnet(pre, x0, x1, Cout):                    # ResnetBlock2D
            Cin = x0.C + (x1.C if x1 else 0)
            slot = self.fc_total
            self.fc_slots[pre + ".time_emb_proj"] = (slot, Cout)
            self.fc_total += Cout
            skip = conv(pre + ".conv_shortcut", x0, x1, Cout, k=1) if Cin != Cout else x0
            a = norm(pre + ".norm1", x0, x1, True)
            h = conv(pre + ".conv1", a, None, Cout, fc=slot)
            conv1_spec = self.specs[-1]
            b = norm(pre + ".norm2", h, None, True)
            self.specs[-1].producer = conv1_spec
            return conv(pre + ".conv2", b, None, Cout, resid=skip)

        def attn(pre, x):                                 # Attention (deprecated-attention-block form), heads of width hd
            C = x.C
            gn = norm(pre + ".group_norm", x, None, False)
            q = conv(pre + ".to_q", gn, None, C, k=1, rshape=(C, C))
            k_ = conv(pre + ".to_k", gn, None, C, k=1, rshape=(C, C))
            v = conv(pre + ".to_v", gn, None, C, k=1, rshape=(C, C))
            o = self._act(pre + ".attn", x.H, x.W, C)
            self.specs.append(_AttnMH(self, q, k_, v, o, C // hd))
            return conv(pre + ".to_out.0", o, None, C, k=1, resid=x, rshape=(C, C))

        self.fc_total = 0
        self.temb_dim = temb
        self.temb_spec = _Temb(self, boc[0], temb, 0, names=("time_embedding.linear_1", "time_embedding.linear_2"), variant=(True, 0.0))
        self.specs.append(self.temb_spec)
        self.x_in = self._act("x_in", self.H, self.W, self.cin_p, needs_grad=False)
        h = conv("conv_in", self.x_in, None, boc[0], rshape=(boc[0], cin, 3, 3))
        hs = [h]
        for i, oc in enumerate(boc):                      # down blocks
            for j in range(lpb):
                h = resnet(f"down_blocks.{i}.resnets.{j}", h, None, oc)
                if cfg["down_attn"][i]:
                    h = attn(f"down_blocks.{i}.attentions.{j}", h)
                hs.append(h)
            if i != len(boc) - 1:
                h = conv(f"down_blocks.{i}.downsamplers.0.conv", h, None, oc, stride=2)
                hs.append(h)
        h = resnet("mid_block.resnets.0", h, None, boc[-1])
        h = attn("mid_block.attentions.0", h)
        h = resnet("mid_block.resnets.1", h, None, boc[-1])
        rev = boc[::-1]
        for i, oc in enumerate(rev):                      # up blocks: layers_per_block + 1 resnets, each fed cat(h, popped skip)
            for j in range(lpb + 1):
                h = resnet(f"up_blocks.{i}.resnets.{j}", h, hs.pop(), oc)
                if cfg["up_attn"][i]:
                    h = attn(f"up_blocks.{i}.attentions.{j}", h)
            if i != len(rev) - 1:
                h = conv(f"up_blocks.{i}.upsamplers.0.conv", h, None, oc, ups=1)
        assert not hs
        h = norm("conv_norm_out", h, None, True)
        self.y_out = conv("conv_out", h, None, self.cout_p, rshape=(cout, boc[0], 3, 3))
        self.temb_spec.fc_total = self.fc_total

    def _default_params(self, seed):
        return default_init_params(self.reference_shapes(), seed)


    def reference_param_order(self):
        """Registration order of diffusers' UNet2DModel as published: conv_in, time_embedding, down_blocks, up_blocks,
        mid_block (assigned after the two ModuleLists), conv_norm_out, conv_out; attentions before resnets inside a block;
        ResnetBlock2D: norm1, conv1, time_emb_proj, norm2, conv2, conv_shortcut; Attention: group_norm, to_q, to_k, to_v,
        to_out.0.  (Unverified offline -- torch optimizer state dicts index parameters by this order.)"""
        top = {"conv_in": 0, "time_embedding": 1, "down_blocks": 2, "up_blocks": 3, "mid_block": 4, "conv_norm_out": 5, "conv_out": 6}
        sub = {"attentions": 0, "resnets": 1, "downsamplers": 2, "upsamplers": 2}
        member = {"norm1": 0, "conv1": 1, "time_emb_proj": 2, "norm2": 3, "conv2": 4, "conv_shortcut": 5,
                  "group_norm": 0, "to_q": 1, "to_k": 2, "to_v": 3, "to_out": 4, "linear_1": 0, "linear_2": 1, "conv": 0}

        def rank(key):
            r = []
            for q in key.split("."):
                if q in top and not r:
                    r.append(top[q])
                elif q.isdigit():
                    r.append(int(q))
                elif q in sub:
                    r.append(sub[q])
                elif q in member:
                    r.append(member[q])
                elif q in ("weight", "bias"):
                    r.append(0 if q == "weight" else 1)
                else:
                    raise KeyError(key)
            return r
        return sorted(self.store.entries, key=rank)


def default_init_params(shapes, seed=0):
    """torch's default nn.Conv2d / nn.Linear / nn.GroupNorm initialisation (what UNet2DModel.from_config leaves in place):
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases, GroupNorm weight 1 / bias 0.  Same distribution, own stream."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    fan = {}
    for k, shp in shapes.items():
        if len(shp) > 1:
            f = 1
            for d in shp[1:]:
                f *= d
            fan[k.rsplit(".", 1)[0]] = f
    for k, shp in shapes.items():
        mod = k.rsplit(".", 1)[0]
        if mod in fan:
            b = 1.0 / math.sqrt(fan[mod])
            out[k] = (torch.rand(shp, generator=g) * 2 - 1) * b
        else:
            out[k] = torch.ones(shp) if k.endswith(".weight") else torch.zeros(shp)
    return out

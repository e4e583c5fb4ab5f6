#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE in the build container.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Needs /root/reference (absent on the GPU box: the fixtures are committed).  Only
inputs/outputs are stored -- never reference source.  Visualization-only
third-party imports of the reference (torchvision, torchmetrics, cv2) are
replaced by inert stand-ins that raise if anything on the math path calls them
(SURVEY.md App. A).
"""
from __future__ import annotations

import argparse
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/code"
sys.path.insert(0, ROOT)

from oracle.unet_ref import random_params  # noqa: E402  (weight recipe shared with the tests)


# --------------------------------------------------------------------------- #
def _stub_modules():
    class _Loud:
        def __init__(self, name):
            self._n = name

        def __call__(self, *a, **k):
            raise RuntimeError(f"stub {self._n} called on the math path")

        def __getattr__(self, k):
            return _Loud(self._n + "." + k)

    def mod(name, attrs=()):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, _Loud(name + "." + a))
        sys.modules[name] = m
        return m

    tv = mod("torchvision", ["transforms", "utils"])
    tv.utils = mod("torchvision.utils", ["save_image", "make_grid"])
    tv.transforms = mod("torchvision.transforms", ["Normalize", "Resize", "RandomHorizontalFlip",
                                                   "RandomVerticalFlip", "Compose"])
    tv.transforms.functional = mod("torchvision.transforms.functional", ["rotate"])
    mod("torchmetrics"); mod("torchmetrics.image")
    mod("torchmetrics.image.fid", ["FrechetInceptionDistance"])
    mod("cv2")
    import matplotlib
    matplotlib.use("Agg")


def seed_all(s=0):
    torch.manual_seed(s); np.random.seed(s); random.seed(s)      # main_train_masked.py:441-445


def base_args(**kw):
    """Defaults of main_train_masked.py:351-417 that the hot path reads."""
    a = argparse.Namespace(
        dir_dataset="synthetic", data_size=8, in_channel=3, out_channel=3, batch_size=4,
        ddpm_num_steps=10, updated_ddpm_num_steps=10, ddpm_schedule="linear", ddpm_schedule_base=10.0,
        scheduler_num_scale_timesteps=1, select_degrade_pixel="thresholding", degrade_channel="1-channel",
        mean_option=0, mean_area="image-wise", shift_type="noise_with_perturbation", noise_mean=0.0,
        sample_latent_shape="zero", sampling="momentum", momentum_adaptive="base_momentum",
        adaptive_momentum_rate=0.9, sampling_mask_dependency="independent", sample_num=2,
        loss_weight_use=False, loss_weight_power_base=10.0, use_ema=False, ema_inv_gamma=1.0, ema_power=0.75,
        ema_max_decay=0.9999, weight_dtype=torch.float32, save_images_epochs=10, mixed_precision="no",
        gradient_accumulation_steps=1)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


# --------------------------------------------------------------------------- #
def gen_schedules(scheduler_mod, out):
    for size in (32, 64):
        for kind in ("linear", "log", "exponential"):
            for T in (10, 50, 250, 1000):
                a = base_args(data_size=size, ddpm_schedule=kind, ddpm_num_steps=T)
                s = scheduler_mod.Scheduler(a)
                steps = s.update_ddpm_num_steps(T)
                key = f"sched_{kind}_{T}_{size}"
                out[key + "_ratio"] = npy(s.get_ratio_list())
                out[key + "_pixels"] = npy(s.get_black_area_num_pixels_all())
                out[key + "_steps"] = np.array(steps)
    # timestep subsets (scheduler.py:173-192)
    for scale in (1, 3):
        a = base_args(data_size=32, ddpm_schedule="linear", ddpm_num_steps=50, scheduler_num_scale_timesteps=scale)
        s = scheduler_mod.Scheduler(a); s.update_ddpm_num_steps(50)
        for epoch in (0, 3, 5, 8):
            out[f"epochsteps_s{scale}_e{epoch}"] = np.array(s.get_timesteps_epoch(epoch, 9))
    # gather + loss weights
    a = base_args(data_size=32, ddpm_schedule="log", ddpm_num_steps=50, select_degrade_pixel="indexing")
    s = scheduler_mod.Scheduler(a); steps = s.update_ddpm_num_steps(50)
    t = torch.tensor([1, 2, steps // 2, steps])
    out["gather_log_idx_t"] = npy(t)
    out["gather_log_idx"] = npy(s.get_black_area_num_pixels_time(t))
    a.select_degrade_pixel = "thresholding"
    out["gather_log_thr"] = npy(s.get_black_area_num_pixels_time(t.float()))
    out["lossw"] = npy(s.get_weight_timesteps(torch.tensor([0, 1, 7, steps - 1]), 10.0))


def gen_degrade(scheduler_mod, out):
    n, c, hw = 4, 3, 8
    g = torch.Generator().manual_seed(11)
    x0 = torch.rand(n, c, hw, hw, generator=g) * 2 - 1
    out["deg_x0"] = npy(x0)
    combos = [
        ("thresholding", "1-channel", "linear", 0, "image-wise"),
        ("thresholding", "1-channel", "linear", 0.5, "image-wise"),
        ("thresholding", "3-channel", "linear", 0, "image-wise"),
        ("thresholding", "1-channel", "exponential", "degraded_area", "image-wise"),
        ("thresholding", "1-channel", "exponential", "degraded_area", "channel-wise"),
        ("thresholding", "3-channel", "log", "non_degraded_area", "channel-wise"),
        ("thresholding", "1-channel", "log", "non_degraded_area", "image-wise"),
        ("indexing", None, "log", 0, "image-wise"),
        ("indexing", None, "log", "degraded_area", "channel-wise"),
    ]
    out["deg_ncombos"] = np.array(len(combos))
    for i, (sel, ch, kind, mo, ma) in enumerate(combos):
        a = base_args(data_size=hw, ddpm_schedule=kind, ddpm_num_steps=10, select_degrade_pixel=sel,
                      degrade_channel=ch, mean_option=mo, mean_area=ma)
        s = scheduler_mod.Scheduler(a); steps = s.update_ddpm_num_steps(10)
        t = torch.tensor([2, steps // 2, steps - 1, steps])[:n]
        seed_all(100 + i)
        amount = s.get_black_area_num_pixels_time(t.float() if sel == "thresholding" else t)
        r = s.degrade_training(amount, x0, mean_option=mo, mean_area=ma)
        seed_all(200 + i)
        r2 = s.degrade_independent_base_sampling(amount[:1].expand(n) if sel == "thresholding" else amount[:1].expand(n),
                                                 x0, mean_option=mo, mean_area=ma)
        r3 = s.degrade_with_mask(x0, r2[1], mo, ma)
        out[f"deg{i}_cfg"] = np.array([str(sel), str(ch), kind, str(mo), ma])
        out[f"deg{i}_t"] = npy(t)
        for j, nm in enumerate(("img", "mask", "dmask", "mean")):
            out[f"deg{i}_train_{nm}"] = npy(r[j])
        for j, nm in enumerate(("img", "mask", "mean")):
            out[f"deg{i}_samp_{nm}"] = npy(r2[j])
        out[f"deg{i}_withmask"] = npy(r3)


def gen_shift(scheduler_mod, out):
    types_ = ["1-d_constant", "3-d_constant", "noise_reduction", "noise_std_reduction",
              "noise_with_perturbation", "non_shift"]
    for (n, hw, tag) in ((4, 8, "n4"), (8, 8, "nEQw")):        # second case: N == W (D10)
        for i, st in enumerate(types_):
            a = base_args(data_size=hw, ddpm_schedule="linear", ddpm_num_steps=10, shift_type=st, noise_mean=0.25)
            s = scheduler_mod.Scheduler(a); s.update_ddpm_num_steps(10)
            t = torch.tensor([(3 * k) % 10 + 1 for k in range(n)]).float()
            seed_all(300 + i)
            sh = s.get_schedule_shift_time(t, torch.zeros(n, 3, hw, hw))
            out[f"shift_{tag}_{st}"] = npy(sh.contiguous())
            out[f"shift_{tag}_{st}_t"] = npy(t)


TINY = dict(in_channels=3, hid_channels=32, out_channels=3, ch_multipliers=[1, 2], num_res_blocks=1,
            apply_attn=[False, True])


def build_ref_unet(unet6, cfg, seed=1234):
    m = unet6.UNet(cfg["in_channels"], cfg["hid_channels"], cfg["out_channels"], cfg["ch_multipliers"],
                   cfg["num_res_blocks"], cfg["apply_attn"])
    p = random_params(cfg, seed)
    assert list(m.state_dict().keys()) == list(p.keys()), "state_dict key grammar mismatch"
    m.load_state_dict(p)
    return m


def gen_unet(unet6, out):
    g = torch.Generator().manual_seed(5)
    # pieces in isolation
    t = torch.tensor([1.0, 7.0, 250.0, 1000.0])
    out["temb_t"] = npy(t); out["temb_128"] = npy(unet6.get_timestep_embedding(t, 128))
    x = torch.randn(2, 32, 8, 8, generator=g)
    pad = unet6.SamePad2d(3, 2)(x)
    out["samepad_x"] = npy(x); out["samepad_shape"] = np.array(pad.shape)
    # full tiny model forward + backward
    m = build_ref_unet(unet6, TINY)
    x = (torch.rand(2, 3, 16, 16, generator=g) * 2 - 1).requires_grad_(True)
    t = torch.tensor([3.0, 41.0])
    y = m(x, t)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    out["unet_x"] = npy(x); out["unet_t"] = npy(t); out["unet_y"] = npy(y); out["unet_gy"] = npy(gy)
    out["unet_gx"] = npy(x.grad)
    sd = dict(m.named_parameters())
    for k in ("in_conv.weight", "out_conv.2.weight", "embed.0.weight", "middle.1.project_in.weight",
              "downsamples.level_0.1.1.weight", "upsamples.level_1.2.1.weight", "upsamples.level_0.0.skip.weight",
              "downsamples.level_1.0.0.fc.weight", "middle.0.norm1.weight", "middle.0.norm1.bias", "out_conv.2.bias"):
        out["unet_grad::" + k] = npy(sd[k].grad)
    # the named preset at 32x32 (Model('unet6',3,32,32,3)): forward only, N=1
    from models import models_Unet
    big = models_Unet.Model("unet6", 3, 32, 32, 3)
    cfg = dict(in_channels=3, hid_channels=128, out_channels=3, ch_multipliers=[1, 2, 2, 2], num_res_blocks=2,
               apply_attn=[False, False, True, False])
    p = random_params(cfg, 77)
    assert list(big.state_dict().keys()) == list(p.keys())
    big.load_state_dict(p)
    xb = torch.rand(1, 3, 32, 32, generator=g) * 2 - 1
    tb = torch.tensor([500.0])
    with torch.no_grad():
        out["unet32_x"] = npy(xb); out["unet32_t"] = npy(tb); out["unet32_y"] = npy(big(xb, tb))
    out["unet32_nparams"] = np.array(sum(v.numel() for v in big.parameters()))


class _Wrap(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, x, t):
        return types.SimpleNamespace(sample=self.net(x, t))


def gen_sampler(scheduler_mod, sampler_mod, unet6, out):
    i = 0
    for dep in ("independent", "dependent_prev"):
        for mode in ("base_momentum", "base_sampling"):
            for sel, ch, kind, st in (("thresholding", "1-channel", "linear", "noise_with_perturbation"),
                                      ("indexing", None, "log", "1-d_constant")):
                a = base_args(data_size=16, ddpm_schedule=kind, ddpm_num_steps=6, select_degrade_pixel=sel,
                              degrade_channel=ch, shift_type=st, sampling_mask_dependency=dep,
                              momentum_adaptive=mode, sample_num=2, sample_latent_shape="uniform")
                s = scheduler_mod.Scheduler(a); steps = s.update_ddpm_num_steps(6)
                ts = s.get_timesteps_epoch(0, 1)
                smp = sampler_mod.Sampler(None, a, s, [None] * 3)
                model = _Wrap(build_ref_unet(unet6, TINY)).eval()
                seed_all(400 + i)
                x0, hist = smp.sample(model, ts)
                out[f"samp{i}_cfg"] = np.array([dep, mode, sel, str(ch), kind, st])
                out[f"samp{i}_ts"] = np.array(ts)
                out[f"samp{i}_x0"] = npy(x0)
                out[f"samp{i}_hist"] = np.stack([npy(h) for h in hist])
                i += 1
    out["samp_n"] = np.array(i)


def gen_sampler_dep_t(scheduler_mod, sampler_mod, unet6, out):
    """`sampling_mask_dependency='dependent_t'` (sampler.py:191-196 -> scheduler.py:480-549) for the sub-case that runs
    upstream: thresholding, 1- or 3-channel masks, mean_option 'degraded_area' or the string "0"."""
    i = 0
    for mode in ("base_momentum", "base_sampling"):
        for ch, kind, st, mo, ma in (("1-channel", "linear", "noise_with_perturbation", "0", "image-wise"),
                                     ("3-channel", "exponential", "noise_reduction", "degraded_area", "channel-wise"),
                                     ("1-channel", "linear", "1-d_constant", "degraded_area", "image-wise")):
            a = base_args(data_size=16, ddpm_schedule=kind, ddpm_num_steps=8, select_degrade_pixel="thresholding",
                          degrade_channel=ch, shift_type=st, sampling_mask_dependency="dependent_t", momentum_adaptive=mode,
                          sample_num=2, sample_latent_shape="uniform", mean_option=mo, mean_area=ma, noise_mean=0.05)
            s = scheduler_mod.Scheduler(a); s.update_ddpm_num_steps(8)
            ts = s.get_timesteps_epoch(0, 1)
            model = _Wrap(build_ref_unet(unet6, TINY)).eval()
            seed_all(800 + i)
            x0, hist = sampler_mod.Sampler(None, a, s, [None] * 3).sample(model, ts)
            out[f"dept{i}_cfg"] = np.array([mode, ch, kind, st, mo, ma])
            out[f"dept{i}_ts"] = np.array(ts)
            out[f"dept{i}_x0"] = npy(x0)
            out[f"dept{i}_hist"] = np.stack([npy(h) for h in hist])
            i += 1
    out["dept_n"] = np.array(i)
    # the combinations that do NOT run upstream: which exception they end in
    for j, (sel, ch, mo) in enumerate((("thresholding", "1-channel", 0), ("thresholding", "1-channel", "non_degraded_area"),
                                       ("indexing", None, "0"))):
        a = base_args(data_size=16, ddpm_schedule="log", ddpm_num_steps=8, select_degrade_pixel=sel, degrade_channel=ch,
                      shift_type="non_shift", sampling_mask_dependency="dependent_t", momentum_adaptive="base_momentum",
                      sample_num=2, sample_latent_shape="zero", mean_option=mo)
        s = scheduler_mod.Scheduler(a); s.update_ddpm_num_steps(8)
        try:
            sampler_mod.Sampler(None, a, s, [None] * 3).sample(_Wrap(build_ref_unet(unet6, TINY)).eval(), s.get_timesteps_epoch(0, 1))
            err = "none"
        except Exception as e:          # noqa: BLE001
            err = type(e).__name__
        out[f"dept_fail{j}"] = np.array([sel, str(ch), str(mo), type(mo).__name__, err])
    out["dept_nfail"] = np.array(3)


def gen_train_step(scheduler_mod, unet6, out):
    """One real `_run_batch` of each trainer with AdamW (main_train_masked.py:134-141)."""
    import accelerate
    import trainer_masked
    import trainer_masked_mean_shift
    tmp = tempfile.mkdtemp()
    dirs = types.SimpleNamespace(list_dir={"train_loss": tmp, "checkpoint": tmp, "ema_sample_img": tmp})
    g = torch.Generator().manual_seed(21)
    x0 = torch.rand(4, 3, 16, 16, generator=g) * 2 - 1
    out["step_x0"] = npy(x0)
    watch = ["in_conv.weight", "out_conv.2.weight", "embed.2.bias", "middle.1.project_out.weight",
             "upsamples.level_0.1.norm2.weight", "in_conv.bias", "downsamples.level_0.0.conv1.weight", "downsamples.level_0.0.fc.weight",
             "downsamples.level_1.0.0.skip.weight", "middle.0.norm1.bias", "upsamples.level_1.0.0.conv2.bias"]
    for name, mod, st, sel, ch, kind, lw in (
            ("ms", trainer_masked_mean_shift, "noise_with_perturbation", "thresholding", "1-channel", "linear", False),
            ("ms_w", trainer_masked_mean_shift, "1-d_constant", "thresholding", "3-channel", "exponential", True),
            ("base", trainer_masked, "non_shift", "indexing", None, "log", False)):
        a = base_args(data_size=16, ddpm_schedule=kind, ddpm_num_steps=10, select_degrade_pixel=sel, degrade_channel=ch,
                      shift_type=st, loss_weight_use=lw, batch_size=4, sample_num=2, sample_latent_shape="zero")
        model = _Wrap(build_ref_unet(unet6, TINY))
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        lr_s = torch.optim.lr_scheduler.LambdaLR(opt, lambda k: 1.0)
        acc = accelerate.Accelerator(cpu=True)
        grads = {}
        orig = acc.clip_grad_norm_

        def rec(params, max_norm, *aa, _m=model, _g=grads, _o=orig, **kk):      # the gradients as they stand BEFORE clipping
            for k, v in _m.net.named_parameters():
                _g[k] = v.grad.detach().clone()
            _g["__norm__"] = _o(_m.parameters(), max_norm, *aa, **kk)
            return _g["__norm__"]
        acc.clip_grad_norm_ = rec
        if name == "base":
            class T(mod.Trainer):                      # constructor bypass
                def __init__(self, args, model, opt, lr_s, acc):
                    self.args, self.model, self.optimizer, self.lr_scheduler, self.accelerator = args, model, opt, lr_s, acc
                    self.ema_model = None; self.lr_list = []; self.global_step = 0
                    self.Scheduler = scheduler_mod.Scheduler(args)
            tr = T(a, model, opt, lr_s, acc)
        else:
            tr = mod.Trainer(a, None, None, [None] * 3, model, None, opt, lr_s, acc)
        steps = tr.Scheduler.update_ddpm_num_steps(a.ddpm_num_steps)
        a.updated_ddpm_num_steps = steps
        tr.timesteps_used_epoch = tr.Scheduler.get_timesteps_epoch(0, 1)
        seed_all(500)
        r = tr._run_batch(0, (x0, None, None), 0, 1, 0, dirs, None)
        loss = r if isinstance(r, float) else r[0]
        out[f"step_{name}_cfg"] = np.array([st, sel, str(ch), kind, str(lw)])
        out[f"step_{name}_loss"] = np.array(loss, dtype=np.float64)
        out[f"step_{name}_pred"] = npy(tr.mask)
        out[f"step_{name}_xin"] = npy(tr.shifted_degrade_img if name != "base" else tr.degraded_img)
        sd = dict(model.net.named_parameters())
        for k in watch:
            out[f"step_{name}_w::{k}"] = npy(sd[k])
        out[f"step_{name}_norm"] = np.array(float(grads.pop("__norm__")))
        for k, v in grads.items():
            out[f"step_{name}_g::{k}"] = npy(v)
        # YARDSTICK: how far the reference's own fp32 gradients are from the exact ones on this fixture = rel-L2 against the
        # oracle's step in fp64 on the same draws.  ('base' holds two fully degraded, all-zero inputs: GroupNorm sees variance
        # exactly 0 there, rstd = 1/sqrt(eps) = 1000 multiplies every rounding error -- 3e-4 against ~1e-6 for the others.)
        from oracle.scheduler_ref import SchedulerRef
        from oracle.trainer_ref import train_step_ref
        from oracle.unet_ref import UNetRef

        class _NoOpt:
            def zero_grad(self): pass
            def step(self): pass
        a64 = argparse.Namespace(**vars(a)); a64.weight_dtype = torch.float64
        s64 = SchedulerRef(a64); s64.update_ddpm_num_steps(a.ddpm_num_steps)
        m64 = UNetRef(TINY, dtype=torch.float64)
        seed_all(500)
        r64 = train_step_ref(m64, _NoOpt(), s64, a64, x0, tr.timesteps_used_epoch, s64.rng, mean_shift=(name != "base"))
        assert np.allclose(npy(r64["x_in"]), out[f"step_{name}_xin"], rtol=0, atol=1e-6), "fp64 run saw other draws"
        sc = min(1.0, 1.0 / (float(r64["grad_norm"]) + 1e-6))                   # undo the in-place clip
        a_ = np.concatenate([(p_.grad / sc).numpy().reshape(-1) for p_ in m64.plist])
        b_ = np.concatenate([npy(grads[k]).reshape(-1).astype(np.float64) for k in m64.keys])
        out[f"step_{name}_g_yardstick"] = np.array(np.linalg.norm(a_ - b_) / np.linalg.norm(b_))
        print(f"  step_{name}: fp32 reference vs fp64 gradients rel-L2 {float(out[f'step_{name}_g_yardstick']):.3e}")


SLICE = dict(in_channels=3, hid_channels=256, out_channels=3, ch_multipliers=[1], num_res_blocks=1,
             apply_attn=[False])            # one level at the preset's width (256 channels): res blocks 256->256, 512->256, attention d=256


def gen_blocks(unet6, out):
    """Pieces in isolation (reference unet6.py:296-362, 257-272) and a preset-WIDTH slice, forward + backward."""
    g = torch.Generator().manual_seed(31)
    # ResidualBlock 32 -> 64 (1x1 skip), 8x8
    rb = unet6.ResidualBlock(32, 64, embed_dim=128)
    for k, v in rb.state_dict().items():
        v.copy_(torch.randn(v.shape, generator=g) * (0.1 if v.dim() > 1 else 0.05) + (1.0 if "norm" in k and k.endswith("weight") else 0.0))
    x = torch.randn(2, 32, 8, 8, generator=g).requires_grad_(True)
    te = torch.randn(2, 128, generator=g).requires_grad_(True)
    y = rb(x, te)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    out["rb_x"] = npy(x); out["rb_temb"] = npy(te); out["rb_y"] = npy(y); out["rb_gy"] = npy(gy)
    out["rb_gx"] = npy(x.grad); out["rb_gtemb"] = npy(te.grad)
    for k, v in rb.named_parameters():
        out["rb_p::" + k] = npy(v); out["rb_g::" + k] = npy(v.grad)
    # AttentionBlock C = 64, 8x8 (L = 64)
    ab = unet6.AttentionBlock(64)
    for k, v in ab.state_dict().items():
        v.copy_(torch.randn(v.shape, generator=g) * (0.15 if v.dim() > 1 else 0.05) + (1.0 if "norm" in k and k.endswith("weight") else 0.0))
    x = torch.randn(2, 64, 8, 8, generator=g).requires_grad_(True)
    y = ab(x)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    out["ab_x"] = npy(x); out["ab_y"] = npy(y); out["ab_gy"] = npy(gy); out["ab_gx"] = npy(x.grad)
    for k, v in ab.named_parameters():
        out["ab_p::" + k] = npy(v); out["ab_g::" + k] = npy(v.grad)
    # SamePad2d(3, 2) + stride-2 conv: VALUES (the downsample of unet6.py:436-440)
    conv = unet6.Conv2d(16, 24, 3, 2)
    for k, v in conv.state_dict().items():
        v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    x = torch.randn(2, 16, 8, 8, generator=g)
    with torch.no_grad():
        out["sp_x"] = npy(x); out["sp_pad"] = npy(unet6.SamePad2d(3, 2)(x)); out["sp_y"] = npy(conv(unet6.SamePad2d(3, 2)(x)))
    for k, v in conv.state_dict().items():
        out["sp_p::" + k] = npy(v)
    # preset-width slice: whole (small) net at hid = 256 on 8x8, all gradients summarised + a subset stored
    m = build_ref_unet(unet6, SLICE, seed=9)
    x = (torch.rand(2, 3, 8, 8, generator=g) * 2 - 1).requires_grad_(True)
    t = torch.tensor([11.0, 640.0])
    y = m(x, t)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    out["slice_x"] = npy(x); out["slice_t"] = npy(t); out["slice_y"] = npy(y); out["slice_gy"] = npy(gy); out["slice_gx"] = npy(x.grad)
    sd = dict(m.named_parameters())
    out["slice_gnorm"] = np.array(float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values()))))
    out["slice_gnorms"] = np.array([float(v.grad.norm()) for v in sd.values()])
    out["slice_keys"] = np.array(list(sd.keys()))
    for k in ("in_conv.weight", "downsamples.level_0.0.conv1.weight", "downsamples.level_0.0.norm2.weight", "middle.1.project_in.weight",
              "middle.1.project_out.bias", "middle.2.conv2.weight", "upsamples.level_0.0.skip.weight", "upsamples.level_0.1.conv1.bias",
              "upsamples.level_0.1.fc.weight", "out_conv.2.weight"):
        gk = sd[k].grad
        out["slice_g::" + k] = npy(gk if gk.numel() < 100000 else gk[:16])       # big filters: the first 16 output channels


def gen_sampler_long(scheduler_mod, sampler_mod, unet6, out):
    """10-step trajectories (all 11 history tensors) and 50-step ones (final sample + every 10th history slot)."""
    i = 0
    for T in (10, 50):
        for dep, mode, sel, ch, kind, st in (("independent", "base_momentum", "thresholding", "1-channel", "linear", "noise_with_perturbation"),
                                             ("dependent_prev", "base_sampling", "thresholding", "3-channel", "exponential", "noise_reduction")):
            a = base_args(data_size=16, ddpm_schedule=kind, ddpm_num_steps=T, select_degrade_pixel=sel, degrade_channel=ch,
                          shift_type=st, sampling_mask_dependency=dep, momentum_adaptive=mode, sample_num=2,
                          sample_latent_shape="normal", noise_mean=0.1)
            s = scheduler_mod.Scheduler(a); s.update_ddpm_num_steps(T)
            ts = s.get_timesteps_epoch(0, 1)
            model = _Wrap(build_ref_unet(unet6, TINY)).eval()
            seed_all(600 + i)
            x0, hist = sampler_mod.Sampler(None, a, s, [None] * 3).sample(model, ts)
            out[f"long{i}_cfg"] = np.array([dep, mode, sel, str(ch), kind, st, str(T)])
            out[f"long{i}_ts"] = np.array(ts)
            out[f"long{i}_x0"] = npy(x0)
            h = np.stack([npy(v) for v in hist])
            out[f"long{i}_hist"] = h if T == 10 else h[:, ::10]
            i += 1
    out["long_n"] = np.array(i)


def _describe(obj, prefix=""):
    """Flat {path: 'kind dtype shape'} description of a nested checkpoint object: the key / shape / dtype GRAMMAR, no values."""
    rows = {}
    if isinstance(obj, torch.Tensor):
        rows[prefix] = f"tensor {str(obj.dtype).replace('torch.', '')} {tuple(obj.shape)}"
    elif isinstance(obj, np.ndarray):
        rows[prefix] = f"ndarray {obj.dtype} {tuple(obj.shape)}"
    elif isinstance(obj, dict):
        if not obj:
            rows[prefix] = "dict empty"
        for k, v in obj.items():
            rows.update(_describe(v, f"{prefix}/{k}" if prefix else str(k)))
    elif isinstance(obj, (list, tuple)):
        if len(obj) > 8 and all(isinstance(v, (int, float)) for v in obj):
            rows[prefix] = f"{type(obj).__name__}[{len(obj)}] of {type(obj[0]).__name__}"
        else:
            if not obj:
                rows[prefix] = f"{type(obj).__name__} empty"
            for i, v in enumerate(obj):
                rows.update(_describe(v, f"{prefix}/{i}"))
    else:
        rows[prefix] = type(obj).__name__
    return rows


def gen_train_traj(scheduler_mod, unet6, out):
    """The reference's `MeanShiftTrainer.train()` END TO END (trainer_masked_mean_shift.py:196-273) on the tiny net: 2 epochs
    x 3 batches of 4 through a real `accelerate.Accelerator(cpu=True)` whose `prepare()` wraps model, optimizer, dataloader
    and LR schedule like main_train_masked.py:299 -- once with gradient_accumulation_steps = 1 and once with 2 (3 batches
    per epoch: the last batch of an epoch syncs on the end-of-dataloader rule).  Stored: per-batch losses, the LR after every
    batch, `timesteps_used_epoch` per epoch, global_step, all weights after training, and the key / shape / dtype
    manifest of the `save_state` directory accelerate wrote (no hooks registered: main_train_masked.py's hooks need
    diffusers, absent here)."""
    import accelerate
    import pickle
    import trainer_masked_mean_shift as mod
    from torch.utils.data import DataLoader, TensorDataset
    g = torch.Generator().manual_seed(51)
    data = torch.rand(12, 3, 16, 16, generator=g) * 2 - 1
    out["traj_data"] = npy(data)
    for gas in (1, 2):
        tmp = tempfile.mkdtemp()
        dirs = types.SimpleNamespace(list_dir={k: os.path.join(tmp, k) for k in ("train_loss", "checkpoint", "ema_sample_img")})
        for d in dirs.list_dir.values():
            os.makedirs(d)
        a = base_args(data_size=16, ddpm_schedule="linear", ddpm_num_steps=10, shift_type="noise_with_perturbation",
                      loss_weight_use=True, batch_size=4, sample_num=2, sample_latent_shape="zero", use_ema=False,
                      scheduler_num_scale_timesteps=2, save_images_epochs=10, gradient_accumulation_steps=gas)
        seed_all(0)                                                        # main_train_masked.py:441-445
        model = _Wrap(build_ref_unet(unet6, TINY))
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        lr_s = torch.optim.lr_scheduler.LambdaLR(opt, lambda k: 1.0 / (1.0 + 0.25 * k))     # every scheduler step is visible
        loader = DataLoader(TensorDataset(data, torch.zeros(12)), batch_size=4, shuffle=False)
        acc = accelerate.Accelerator(cpu=True, gradient_accumulation_steps=gas)
        model, opt, loader, lr_s = acc.prepare(model, opt, loader, lr_s)
        tr = mod.Trainer(a, loader, None, [None] * 3, model, None, opt, lr_s, acc)
        losses, used, ts_batch, syncs = [], [], [], []
        orig_batch, orig_epoch = tr._run_batch, tr._run_epoch

        def run_batch(*aa, **kk):
            r = orig_batch(*aa, **kk)
            losses.append(r); syncs.append(bool(acc.sync_gradients))
            return r

        def run_epoch(*aa, **kk):
            r = orig_epoch(*aa, **kk)
            used.append(list(tr.timesteps_used_epoch))
            return r
        tr._run_batch, tr._run_epoch = run_batch, run_epoch
        seed_all(900 + gas)
        tr.train(0, 2, 0, 0, dirs, None)
        tag = f"traj_g{gas}"
        out[tag + "_losses"] = np.array(losses, dtype=np.float64)
        out[tag + "_sync"] = np.array(syncs)
        out[tag + "_lr"] = np.array(tr.lr_list, dtype=np.float64)
        out[tag + "_global_step"] = np.array(tr.global_step)
        out[tag + "_used_e0"] = np.array(used[0]); out[tag + "_used_e1"] = np.array(used[1])
        net = acc.unwrap_model(model).net
        for k, v in net.named_parameters():
            out[tag + "_w::" + k] = npy(v)
        # ---- what accelerate's save_state wrote (trainer_masked_mean_shift.py:267-269)
        ck = os.path.join(dirs.list_dir["checkpoint"], "checkpoint-epoch-1")
        files = sorted(os.listdir(ck))
        out[tag + "_ckpt_dirs"] = np.array(sorted(os.listdir(dirs.list_dir["checkpoint"])))
        out[tag + "_ckpt_files"] = np.array(files)
        rows = {}
        for f in files:
            path = os.path.join(ck, f)
            if f.endswith(".safetensors"):
                from safetensors.torch import load_file
                rows.update(_describe(load_file(path), f))
            elif f.endswith(".bin") or f.endswith(".pkl"):        # (random_states_<rank>.pkl is a torch.save file too)
                rows.update(_describe(torch.load(path, map_location="cpu", weights_only=False), f))
        keys = sorted(rows)
        out[tag + "_ckpt_manifest"] = np.array([f"{k} :: {rows[k]}" for k in keys])
        if gas == 1:           # the optimizer state values, to check a load of OUR optimizer.bin against (exp_avg of two tensors + step)
            osd = torch.load(os.path.join(ck, "optimizer.bin"), map_location="cpu", weights_only=False)
            out["traj_g1_opt_step"] = np.array(float(osd["state"][0]["step"]))
            out["traj_g1_opt_exp_avg_0"] = npy(osd["state"][0]["exp_avg"])
            out["traj_g1_opt_exp_avg_sq_5"] = npy(osd["state"][5]["exp_avg_sq"])
            out["traj_g1_opt_group"] = np.array([f"{k}={v}" for k, v in osd["param_groups"][0].items() if k != "params"])
            ssd = torch.load(os.path.join(ck, "scheduler.bin"), map_location="cpu", weights_only=False)
            out["traj_g1_sched"] = np.array([f"{k}={v}" for k, v in ssd.items()])
        print(f"  traj gas={gas}: losses {np.round(losses, 4).tolist()} sync {syncs} lr {np.round(tr.lr_list, 6).tolist()} "
              f"global_step {tr.global_step} files {files}")


def gen_train_grads(scheduler_mod, unet6, out):
    """The gradient tensors of one real mean-shift `_run_batch` (before clipping) and the clipping norm."""
    import accelerate
    import trainer_masked_mean_shift as mod
    tmp = tempfile.mkdtemp()
    dirs = types.SimpleNamespace(list_dir={"train_loss": tmp, "checkpoint": tmp, "ema_sample_img": tmp})
    g = torch.Generator().manual_seed(23)
    x0 = torch.rand(4, 3, 16, 16, generator=g) * 2 - 1
    a = base_args(data_size=16, ddpm_schedule="linear", ddpm_num_steps=10, shift_type="noise_with_perturbation", loss_weight_use=True,
                  batch_size=4, sample_num=2, sample_latent_shape="zero")
    model = _Wrap(build_ref_unet(unet6, TINY))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    lr_s = torch.optim.lr_scheduler.LambdaLR(opt, lambda k: 1.0)
    acc = accelerate.Accelerator(cpu=True)
    grads = {}
    orig = acc.clip_grad_norm_

    def rec(params, max_norm, *aa, **kk):
        for k, v in model.net.named_parameters():
            grads[k] = v.grad.detach().clone()
        grads["__norm__"] = orig(model.parameters(), max_norm, *aa, **kk)
        return grads["__norm__"]
    acc.clip_grad_norm_ = rec
    tr = mod.Trainer(a, None, None, [None] * 3, model, None, opt, lr_s, acc)
    a.updated_ddpm_num_steps = tr.Scheduler.update_ddpm_num_steps(a.ddpm_num_steps)
    tr.timesteps_used_epoch = tr.Scheduler.get_timesteps_epoch(0, 1)
    seed_all(501)
    loss = tr._run_batch(0, (x0, None, None), 0, 1, 0, dirs, None)
    out["tg_x0"] = npy(x0); out["tg_loss"] = np.array(loss, dtype=np.float64); out["tg_norm"] = np.array(float(grads.pop("__norm__")))
    out["tg_xin"] = npy(tr.shifted_degrade_img)
    for k, v in grads.items():
        out["tg_g::" + k] = npy(v)


def gen_evaluate(scheduler_mod, sampler_mod, out):
    """The evaluation caller (reference tester.py / utils/datautils.py / sampler.py:46-83 'data' branch) on small inputs."""
    import tester as tester_mod
    from utils.datautils import normalize01
    from oracle.evaluate_ref import data_mean_histogram
    g = torch.Generator().manual_seed(41)
    data = torch.rand(20, 3, 8, 8, generator=g) * 2 - 1
    batch = torch.rand(12, 3, 8, 8, generator=g)
    batch[3] = batch[0] + 0.02 * torch.randn(3, 8, 8, generator=g)        # near-duplicates inside the batch
    batch[7] = batch[5] * 1.5
    batch[9] = normalize01(data[4:5])[0] + 0.01 * torch.randn(3, 8, 8, generator=g)     # close to a data image
    prev = torch.cat([batch[1:2] + 0.01 * torch.randn(1, 3, 8, 8, generator=g), torch.rand(3, 3, 8, 8, generator=g)])
    const = torch.full((1, 3, 8, 8), 0.3)                                  # a constant image: normalize01 -> NaN -> 0
    out["ev_data"] = npy(data); out["ev_batch"] = npy(batch); out["ev_prev"] = npy(prev)
    out["ev_norm01"] = npy(normalize01(torch.cat([data[:3], const])))
    ds = [(data[i], 0) for i in range(len(data))]
    me = types.SimpleNamespace(cosine_similarity_th=0.9, dataset=ds, args=types.SimpleNamespace(sample_num=7, data_size=8))
    T_ = tester_mod.Tester
    for nm in ("cosine_similarity", "_compute_similarity"):
        setattr(me, nm, getattr(T_, nm).__get__(me))
    out["ev_sim"] = npy(T_._compute_similarity(me, batch, normalize01(data)))
    uniq = T_.remove_duplicates_in_batches(me, batch)
    out["ev_unique_in_batch"] = npy(uniq)
    out["ev_unique_across"] = npy(T_.remove_duplicates_across_batches(me, uniq, prev))
    out["ev_nn_idx"] = npy(T_.get_nearest_neighbor_idx(me, batch))
    # 'data' initial latent (sampler.py:46-69) from a data-mean histogram (main_train_masked.py:60-87 restated in oracle/)
    for area in ("image-wise", "channel-wise"):
        a = base_args(data_size=8, sample_num=6, sample_latent_shape="data", mean_area=area)
        s = scheduler_mod.Scheduler(a)
        hist = data_mean_histogram(data, 6, area)
        smp = sampler_mod.Sampler(None, a, s, hist)
        seed_all(700)
        out[f"ev_latent_{area}"] = npy(smp._get_latent_initial(None).contiguous())
        out[f"ev_hist_cum_{area}"] = npy(hist[2])


def main():
    _stub_modules()
    sys.path.insert(0, REF)
    import scheduler as scheduler_mod
    import sampler as sampler_mod
    from models.unet import unet6
    torch.set_num_threads(4)
    jobs = {
        "schedules": lambda o: gen_schedules(scheduler_mod, o),
        "degrade": lambda o: gen_degrade(scheduler_mod, o),
        "shift": lambda o: gen_shift(scheduler_mod, o),
        "unet": lambda o: gen_unet(unet6, o),
        "sampler": lambda o: gen_sampler(scheduler_mod, sampler_mod, unet6, o),
        "train_step": lambda o: gen_train_step(scheduler_mod, unet6, o),
        "blocks": lambda o: gen_blocks(unet6, o),
        "sampler_long": lambda o: gen_sampler_long(scheduler_mod, sampler_mod, unet6, o),
        "train_grads": lambda o: gen_train_grads(scheduler_mod, unet6, o),
        "sampler_dep_t": lambda o: gen_sampler_dep_t(scheduler_mod, sampler_mod, unet6, o),
        "train_traj": lambda o: gen_train_traj(scheduler_mod, unet6, o),
        "evaluate": lambda o: gen_evaluate(scheduler_mod, sampler_mod, o),
    }
    only = sys.argv[1:]
    for name, fn in jobs.items():
        if only and name not in only:
            continue
        o = {}
        fn(o)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **o)
        print(f"{name}: {len(o)} arrays -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()

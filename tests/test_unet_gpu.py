"""GPU parity of the assembled HIP U-Net (forward and full backward) against the
golden vectors of the reference and against the CPU oracle's autograd."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden.make_golden import TINY  # noqa: E402


def _rel(a, b):
    a = torch.as_tensor(a).float().cpu()
    b = torch.as_tensor(b).float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def _run_tiny(dt, golden):
    from mdm import unet as U
    from mdm import ops
    from oracle.unet_ref import random_params
    g = golden("unet")
    x, t, gy = (torch.from_numpy(g[k]) for k in ("unet_x", "unet_t", "unet_gy"))
    net = U.UNet(TINY, N=2, H=16, W=16, dtype=dt, params=random_params(TINY))
    y = net(x, t).sample
    # backward: hand dL/dy to the net as NHWC (padded channels zero)
    dev = net.device
    net.zero_grad()
    ops.nchw_to_nhwc(dt, gy.to(dev), net.y_out.grad, 2, 3, 16, 16, net.cout_p)
    net.run_backward()
    torch.cuda.synchronize()
    return net, y.cpu(), net.store.grad_dict(), g


@pytest.mark.parametrize("dt,tol_y,tol_g", [(0, 2e-4, 2e-3), (1, 3e-2, 8e-2)])
def test_unet_tiny_forward_backward(golden, dt, tol_y, tol_g):
    net, y, grads, g = _run_tiny(dt, golden)
    assert _rel(y, g["unet_y"]) < tol_y
    # every parameter gradient against the oracle's autograd (the oracle itself is pinned to the
    # reference on the subset stored in the fixture, tests/test_oracle_golden.py)
    from oracle.unet_ref import UNetRef
    m = UNetRef(TINY)
    xo = torch.from_numpy(g["unet_x"]).requires_grad_(True)
    yo = m(xo, torch.from_numpy(g["unet_t"])).sample
    (yo * torch.from_numpy(g["unet_gy"])).sum().backward()
    want = {k: p.grad for k, p in zip(m.keys, m.plist)}
    assert set(want) == set(grads)
    # Some gradients are mathematically zero (a bias that feeds a GroupNorm with one channel per
    # group): both sides then hold rounding noise, so the denominator gets a floor tied to the
    # typical gradient magnitude per element.
    rms = float(torch.cat([w.reshape(-1) for w in want.values()]).pow(2).mean().sqrt())

    def err(a, b):
        a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
        return float((a - b).norm() / (b.norm() + 1e-2 * rms * b.numel() ** 0.5))
    # in bf16 the noise on those zero gradients is bf16-sized: per-tensor check only where the oracle's
    # gradient is not negligible next to its peers (they all stay in the global check below)
    med = sorted(float(w.norm()) for w in want.values())[len(want) // 2]
    keys = [k for k in want if dt == 0 or float(want[k].norm()) > 1e-2 * med]
    assert len(keys) > 0.8 * len(want)
    worst = max((err(grads[k], want[k]), k) for k in keys)
    allg = torch.cat([grads[k].reshape(-1) for k in want]), torch.cat([want[k].reshape(-1) for k in want])
    from _notes import note
    note("unet_tiny", dict(dtype=dt, rel_l2_y=_rel(y, g["unet_y"]), rel_l2_grads=_rel(*allg), worst_tensor=worst[0], which=worst[1]))
    assert worst[0] < tol_g, worst
    assert _rel(*allg) < tol_g / 2
    for k in g.files:
        if k.startswith("unet_grad::"):
            assert err(grads[k.split("::")[1]], g[k]) < tol_g, k


@pytest.mark.parametrize("dt,tol", [(0, 2e-4), (1, 3e-2)])
def test_unet32_preset_forward(golden, dt, tol):
    from mdm import unet as U
    from oracle.unet_ref import random_params
    g = golden("unet")
    cfg = U.unet6_config(32)
    net = U.UNet(cfg, N=1, H=32, W=32, dtype=dt, params=random_params(cfg, 77))
    assert net.num_parameters() == int(g["unet32_nparams"])
    y = net(torch.from_numpy(g["unet32_x"]), torch.from_numpy(g["unet32_t"])).sample
    torch.cuda.synchronize()
    assert _rel(y, g["unet32_y"]) < tol


@pytest.mark.parametrize("n", [1, 5])
def test_unet32_preset_forward_split_products(golden, n):
    """UNet(dtype=F32, f32_products='split'): fp32 storage, the convolutions' products as bf16 hi / lo pairs on the bf16 matrix pipe
    (mdm_gemm_desc.B_split / f32_split).  The 35.75 M preset against the reference's own forward (the fixture) and against the exact
    fp32 plan over the SAME parameter store: 5e-5 (measured 1.2e-5 / 1.5e-5 through the ~60 convolutions of the net, ~2e-6 per layer; bf16
    storage: 1e-2; no atomics anywhere on this path, so the figure is the same on every box).  A weight update must reach the split shadow."""
    from mdm import unet as U
    from oracle.unet_ref import random_params
    g = golden("unet")
    cfg = U.unet6_config(32)
    exact = U.UNet(cfg, N=n, H=32, W=32, dtype=0, params=random_params(cfg, 77)).eval()
    split = U.UNet(cfg, N=n, H=32, W=32, dtype=0, store=exact.store, f32_products="split").eval()
    x = torch.from_numpy(g["unet32_x"]).repeat(n, 1, 1, 1)
    x = x + 0.25 * torch.arange(n).view(n, 1, 1, 1)
    t = torch.from_numpy(g["unet32_t"]).repeat(n) + 7 * torch.arange(n)
    ye = exact(x, t).sample.clone()
    ys = split(x, t).sample.clone()
    torch.cuda.synchronize()
    assert _rel(ye[:1], g["unet32_y"]) < 2e-4 and _rel(ys[:1], g["unet32_y"]) < 2e-4
    assert 0.0 < _rel(ys, ye.cpu()) < 5e-5, _rel(ys, ye.cpu())
    # the shadow follows the weights: scale one filter through the store's own interface and compare again
    sd = exact.state_dict()
    sd["downsamples.level_1.0.conv1.weight"] = sd["downsamples.level_1.0.conv1.weight"] * 1.5
    exact.load_state_dict(sd)
    ye2 = exact(x, t).sample.clone()
    ys2 = split(x, t).sample.clone()
    torch.cuda.synchronize()
    assert _rel(ye2, ye.cpu()) > 1e-3
    assert _rel(ys2, ye2.cpu()) < 5e-5


@pytest.mark.parametrize("dt", [0, 1])
def test_sampling_plan_follows_the_models_weights(dt):
    """UNet.sampling_plan: the reverse sampler's plan (fp32 storage, split products by default) carries the model's CURRENT fp32 master
    weights -- shared store for an fp32 model, a device copy into a second fp32 store for a bf16 model -- also after they change."""
    from mdm import unet as U
    from oracle.unet_ref import random_params
    cfg = dict(TINY, hid_channels=64)
    g = torch.Generator().manual_seed(11)
    x, t = torch.randn(3, 3, 16, 16, generator=g), torch.tensor([3.0, 400.0, 999.0])
    model = U.UNet(cfg, N=2, H=16, W=16, dtype=dt, params=random_params(cfg, 21))
    for seed in (21, 22):
        if seed != 21:
            model.load_state_dict(random_params(cfg, seed))
        want = U.UNet(cfg, N=3, H=16, W=16, dtype=0, params=model.state_dict()).eval()(x, t).sample.clone()
        for prec, tol in (("f32_split", 5e-5), ("f32", 2e-6)):
            plan = model.sampling_plan(3, prec).eval()
            assert plan.dt == 0 and plan.N == 3 and plan.split_products == (prec == "f32_split")
            got = plan(x, t).sample.clone()
            torch.cuda.synchronize()
            assert _rel(got, want.cpu()) < tol, (seed, prec, _rel(got, want.cpu()))
    assert model.sampling_plan(2, "model") is model


def test_state_dict_roundtrip_and_fresh_init():
    from mdm import unet as U
    from oracle.unet_ref import param_shapes, random_params
    p = random_params(TINY, 5)
    net = U.UNet(TINY, N=1, H=16, W=16, dtype=1, params=p)
    sd = net.state_dict()
    assert list(param_shapes(TINY)) and set(sd) == set(p)
    for k in p:
        assert torch.equal(sd[k], p[k]), k
    fresh = U.UNet(TINY, N=1, H=16, W=16, dtype=1)
    sd = fresh.state_dict()
    assert float(sd["out_conv.2.weight"].abs().max()) < 1e-4 and float(sd["in_conv.weight"].abs().max()) > 1e-2
    assert float(sd["middle.0.norm1.weight"].min()) == 1.0


# ------------------------------------------------------------------------------- other BASELINE configs
CFG4_TINY = dict(in_channels=4, hid_channels=32, out_channels=4, ch_multipliers=[1, 2, 2], num_res_blocks=1,
                 apply_attn=[True, True, True])          # cfg4: C=4, attention at every level (L = 1024, 256, 64)
CFG3_TINY = dict(in_channels=3, hid_channels=32, out_channels=3, ch_multipliers=[1, 2, 2, 2], num_res_blocks=2,
                 apply_attn=[False, False, True, False])  # cfg3: the unet6 preset's topology at 64x64, narrow
CFG_WIDE = dict(in_channels=3, hid_channels=128, out_channels=3, ch_multipliers=[1, 2, 2], num_res_blocks=1,
                apply_attn=[False, True, False])         # the preset's widths on 16x16: 8x8 and 4x4 levels with >= 4 channels per
                                                         # group -> whole-image halo tiles, GroupNorm backward fused into the data gradient


CFG4_FULL = dict(in_channels=4, hid_channels=128, out_channels=4, ch_multipliers=[1, 2, 2, 2], num_res_blocks=2,
                 apply_attn=[True, True, True, True])     # BASELINE cfg4 at FULL width: 38.72 M parameters, attention L = 1024 (d 128), 256, 64, 16 (d 256)
CFG3_FULL = dict(in_channels=3, hid_channels=128, out_channels=3, ch_multipliers=[1, 2, 2, 2], num_res_blocks=2,
                 apply_attn=[False, False, True, False])  # BASELINE cfg3: the unet6 preset on 64x64 (GroupNorm over 16 384-element groups, L = 256)


@pytest.mark.parametrize("name,cfg,hw,n", [("cfg4_attn_everywhere_C4", CFG4_TINY, 32, 2), ("cfg3_64x64", CFG3_TINY, 64, 2),
                                           ("preset_widths_16x16", CFG_WIDE, 16, 4), ("cfg4_full_width", CFG4_FULL, 32, 2),
                                           ("cfg3_full_width_64x64", CFG3_FULL, 64, 2)])
# bf16 bars at <= 3x the measured values (profiles/r04_parity_notes.jsonl: y <= 1.4e-2, all gradients <= 2.2e-2; the worst single tensor is
# 0.09 on the 4-channel cfg4 model -- tiny tensors --, <= 0.04 on the others)
@pytest.mark.parametrize("dt,tol_y,tol_g", [(0, 3e-4, 3e-3), (1, 4e-2, 6e-2)])
def test_other_configs_forward_backward_vs_oracle(name, cfg, hw, n, dt, tol_y, tol_g):
    from mdm import ops
    from mdm import unet as U
    from oracle.unet_ref import UNetRef, random_params
    p = random_params(cfg, 11)
    g = torch.Generator().manual_seed(13)
    c = cfg["in_channels"]
    x = torch.rand(n, c, hw, hw, generator=g) * 2 - 1
    t = torch.tensor([5.0, 321.0, 77.0, 950.0][:n])
    gy = torch.randn(n, c, hw, hw, generator=g)
    net = U.UNet(cfg, N=n, H=hw, W=hw, dtype=dt, params=p)
    y = net(x, t).sample
    net.zero_grad()
    ops.nchw_to_nhwc(dt, gy.to(net.device), net.y_out.grad, n, c, hw, hw, net.cout_p)
    net.run_backward()
    torch.cuda.synchronize()
    grads = net.store.grad_dict()
    m = UNetRef(cfg, p)
    yo = m(x, t).sample
    (yo * gy).sum().backward()
    assert _rel(y, yo.detach()) < tol_y
    want = {k: q.grad for k, q in zip(m.keys, m.plist)}
    a = torch.cat([grads[k].reshape(-1) for k in want])
    b = torch.cat([want[k].reshape(-1) for k in want])
    assert _rel(a, b) < tol_g
    med = sorted(float(w.norm()) for w in want.values())[len(want) // 2]
    worst = max((_rel(grads[k], want[k]), k) for k in want if float(want[k].norm()) > 1e-2 * med)
    from _notes import note
    note("other_configs", dict(name=name, dtype=dt, rel_l2_y=_rel(y, yo.detach()), rel_l2_grads=_rel(a, b), worst_tensor=worst[0], which=worst[1]))
    assert worst[0] < (2 * tol_g if dt == 0 else (0.25 if name == "cfg4_attn_everywhere_C4" else 0.12)), worst


@pytest.mark.parametrize("dt,tol_y,tol_g", [(0, 3e-4, 3e-3), (1, 2.5e-2, 1e-1)])      # bf16 y: 8.5e-3 measured
def test_preset_width_slice_vs_reference(golden, dt, tol_y, tol_g):
    """A one-level net at the preset's width (hid 256: res blocks 256->256 and 512->256, attention d = 256, 8x8 maps):
    forward, input gradient, per-tensor gradient norms and a stored subset of gradients of the REFERENCE itself."""
    from golden.make_golden import SLICE
    from mdm import ops
    from mdm import unet as U
    from oracle.unet_ref import random_params
    g = golden("blocks")
    x, t, gy = (torch.from_numpy(g[k]) for k in ("slice_x", "slice_t", "slice_gy"))
    net = U.UNet(SLICE, N=2, H=8, W=8, dtype=dt, params=random_params(SLICE, 9))
    y = net(x, t).sample
    net.zero_grad()
    ops.nchw_to_nhwc(dt, gy.to(net.device), net.y_out.grad, 2, 3, 8, 8, net.cout_p)
    net.run_backward()
    torch.cuda.synchronize()
    assert _rel(y, g["slice_y"]) < tol_y
    from _notes import note
    note("preset_width_slice", dict(dtype=dt, rel_l2_y=_rel(y, g["slice_y"])))
    grads = net.store.grad_dict()
    keys = [str(k) for k in g["slice_keys"]]
    assert set(keys) == set(grads)
    norms = np.array([float(grads[k].norm()) for k in keys])
    want = g["slice_gnorms"]
    big = want > 1e-2 * np.median(want)
    assert np.allclose(norms[big], want[big], rtol=5 * tol_g), np.abs(norms[big] / want[big] - 1).max()
    for k in g.files:
        if k.startswith("slice_g::"):
            w = g[k]
            assert _rel(grads[k.split("::")[1]][:w.shape[0]], w) < 2 * tol_g, k

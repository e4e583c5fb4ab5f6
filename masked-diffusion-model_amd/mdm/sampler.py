"""Reverse (cold-diffusion) sampler on the GPU -- same surface as the reference `Sampler`.

Mirrors reference code/sampler.py: `_get_latent_initial` (:46-83), `sample` (:102-106) and
`_sample_mean_shift_momentum` (:109-261) for the options that run upstream at HEAD
(`sampling_mask_dependency` in {independent, dependent_prev, dependent_t -- the last for
'thresholding' with mean_option 'degraded_area' or "0", scheduler.py:480-549}, `momentum_adaptive`
in {base_sampling, base_momentum}).

What is different from the reference loop (same arithmetic, same RNG order in replay mode):
  * the whole state stays on the GPU; nothing is copied to the host inside the loop.  The
    reference's 11 history tensors are optional (`args.sample_history`: True -> returned on
    the CPU like the reference, "device" -> kept in HBM, False -> skipped, empty list);
  * per step: one fused shift+perturb kernel, the U-Net forward plan, one fused
    x0-reconstruction kernel, two degrade kernels and one update kernel.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops
from ._lib import call, ptr, stream

HISTORY_NAMES = ["sample_t", "shift", "shifted", "mask", "shifted_result", "sample_0",
                 "degraded_mask", "degraded_mask_next", "degraded_t", "difference", "degraded_next_t"]


def shard_bounds(n, rank, world):
    """[lo, hi) of rank's share when n independent samples are dealt to `world` ranks (the first n % world ranks
    take one more).  Samples are independent (sampler.py:137-258), so the split needs no collective."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_shards(local, n, group=None):
    """All-gather of uneven shards along dim 0 -> the full [n, ...] tensor on every rank (the ONE collective of a
    sharded sampling run, after the loop).  Works on any backend: shards are padded to the largest one."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)]
    big = max(sizes)
    pad = torch.zeros((big,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:k] for p, k in zip(parts, sizes)], 0)


class Sampler:
    def __init__(self, dataset, args, Scheduler, dataset_hist):
        self.dataset, self.args, self.Scheduler, self.dataset_hist = dataset, args, Scheduler, dataset_hist
        # Optional observer `f(i, slot, x_t)` called at the top of every reverse step of the host-driven loop; it may
        # overwrite x_t in place.  The parity tests use it for TEACHER FORCING (x_t <- the oracle's x_t of that step): an
        # untrained U-Net iterated 1000 times is a chaotic map -- the reference's own fp32 run is O(1) away from its fp64
        # run by then -- so a long free-running comparison measures conditioning; the per-step one measures the kernels.
        self.step_hook = None

    # ---- multi-GPU: `sample_num` is partitioned over the ranks of the default process group ----------------
    def _shard(self):
        """(rank, world) when sharded sampling applies, else (0, 1).  Off with `args.shard_sampling=False`."""
        import torch.distributed as dist
        if not getattr(self.args, "shard_sampling", True) or not (dist.is_available() and dist.is_initialized()):
            return 0, 1
        return dist.get_rank(), dist.get_world_size()

    def local_sample_num(self):
        rank, world = self._shard()
        lo, hi = shard_bounds(self.args.sample_num, rank, world)
        return hi - lo

    def _get_latent_initial(self, model=None):
        """Constant-colour start image per sample, drawn on the host like sampler.py:46-83.  Sharded sampling: drawn for the
        FULL sample_num on every rank (all ranks are seeded alike, main_train_masked.py:441-445) and sliced to this rank's rows
        by the caller -- a rank drawing only its own share would start from the same colours as every other rank."""
        a = self.args
        if a.mean_area == "image-wise":
            d = 1
        elif a.mean_area == "channel-wise":
            d = 3
        else:
            raise UnboundLocalError("mean_area")
        shape = a.sample_latent_shape.lower()
        if shape == "data":
            hshape, edges, cum = self.dataset_hist
            idx = torch.searchsorted(cum, torch.rand(a.sample_num))
            idx = np.unravel_index(idx, hshape)
            mean = torch.empty(a.sample_num, 0)
            for c in range(d):
                r = torch.rand(a.sample_num)
                lo, hi = edges[c][idx[c]], edges[c][idx[c] + 1]
                mean = torch.cat((mean, ((hi - lo) * r + lo).unsqueeze(-1)), 1)
        elif shape == "zero":
            mean = torch.zeros(a.sample_num, d)
        elif shape == "normal":
            mean = torch.randn(a.sample_num, d)
        elif shape == "uniform":
            mean = torch.empty(a.sample_num, d).uniform_(-1, 1)
        else:
            raise IndexError(f"sample_latent_shape={shape!r} does not run upstream")
        return mean[:, :, None, None].expand(a.sample_num, a.out_channel, a.data_size, a.data_size)

    def sample(self, model, timesteps_used_epoch, interpolation_shift=None):
        """sampler.py:102-106.  With a process group of W ranks every rank runs the loop on its own
        `local_sample_num()` samples (`model` must be built for that batch; per-rank Philox key, see
        Scheduler) and the shards are all-gathered once at the end: every rank returns [sample_num, C, H, W]."""
        rank, world = self._shard()
        if world == 1:
            return self._sample_mean_shift_momentum(model, timesteps_used_epoch)
        import argparse
        full = self.args
        n = full.sample_num
        lo, hi = shard_bounds(n, rank, world)
        try:
            if hi == lo:
                raise ValueError(f"sample_num={n} < world size {world}: a rank would sample nothing")
            latent = self._get_latent_initial(model)[lo:hi]           # the full draw, this rank's rows
            self.args = argparse.Namespace(**vars(full))
            self.args.sample_num = hi - lo
            self.Scheduler.replay_rows = (lo, hi, n)                  # every later host draw: full batch, sliced (replay mode)
            x0, hist = self._sample_mean_shift_momentum(model, timesteps_used_epoch, latent=latent)
        finally:
            self.args = full
            self.Scheduler.replay_rows = None
        def gather_hist(h):             # [T+1, n_local, C, H, W], on the host when args.sample_history is True
            g = gather_shards(h.to(x0.device).transpose(0, 1).contiguous(), n).transpose(0, 1)
            return g if h.is_cuda else g.cpu()
        return gather_shards(x0, n), [gather_hist(h) for h in hist]

    # ------------------------------------------------------------------------------
    def _predict(self, model, x_in_nchw, time, fused_nhwc):
        """-> (pred as NHWC tensor, Cp, dtype) ready for mdm_sampler_x0."""
        from .unet import UNet
        if isinstance(model, UNet):
            if not fused_nhwc:      # x_in was not written into the net by the shift kernel
                ops.nchw_to_nhwc(model.dt, x_in_nchw, model.x_in.data, model.N, model.cin, model.H, model.W, model.cin_p)
            model.t_in.copy_(time, non_blocking=True)
            model.run_forward()
            return model.y_out.data, model.cout_p, model.dt
        pred = model(x_in_nchw, time).sample.to(x_in_nchw.device, torch.float32).contiguous()   # any callable model
        n, c, h, w = pred.shape
        cp = (c + 7) // 8 * 8
        nh = torch.empty(n, h, w, cp, device=pred.device)
        ops.nchw_to_nhwc(_lib.F32, pred, nh, n, c, h, w, cp)
        return nh, cp, _lib.F32

    def _sample_graph(self, model, timesteps, x_t, dep, mode):
        """The same loop as `_sample_mean_shift_momentum` (history off, device RNG) with ONE hipGraph per reverse step:
        the per-step scalars (t, t-1, shift ratio, degrade amounts, Philox offset) are produced on the device by
        `mdm_sampler_step_params` from a step counter, so T replays need no host work in between."""
        from .scheduler import SHIFT_KINDS, _fill_mode
        a, S = self.args, self.Scheduler
        dev = S.device
        T = len(timesteps)
        n, c, hw = a.sample_num, a.out_channel, a.data_size
        HW = hw * hw
        st = a.shift_type
        if st not in SHIFT_KINDS:
            raise UnboundLocalError(f"shift_time undefined for shift_type={st!r}")
        kind = SHIFT_KINDS[st]
        per_col = int(S.reference_quirks and kind in (3, 4) and n == hw and n > 1)
        sel = a.select_degrade_pixel
        fm, fc = _fill_mode(a.mean_option, a.mean_area)
        Cm = S._check_degrade_args(x_t)
        amount_tab = (S.pixels_dev if sel == "indexing" else S.ratio_dev).to(dev, torch.float64).contiguous()
        ratio_tab = S.ratio_dev.to(dev, torch.float64).contiguous()
        ts_dev = torch.as_tensor([int(t) for t in timesteps], dtype=torch.int32, device=dev)
        ctr = torch.zeros(1, dtype=torch.int32, device=dev)
        f64 = lambda: torch.empty(n, dtype=torch.float64, device=dev)
        ratio, amt_t, amt_next = f64(), f64(), f64()
        img = lambda: torch.empty_like(x_t)
        s, x_in, x0_hat, d_t, d_next, m_t, diff = img(), img(), img(), img(), img(), img(), torch.zeros_like(x_t)
        m_next = torch.zeros(n, c, hw, hw, device=dev)
        mi_t, mi_next = (img(), img()) if sel == "indexing" else (None, None)
        mp = torch.empty(n, c, device=dev)
        rng = S.dev_rng.dev

        def degrade(amount, stream_id, mask_src, out, mask_out, mi):
            if mask_src is not None:                      # degrade_with_mask (scheduler.py:572-598)
                call("mdm_degrade", ptr(x0_hat), None, ptr(mask_src), None, 1, ptr(rng), 0, n, c, HW, c, fm, fc, ptr(out), None, ptr(mp), stream())
            elif sel == "indexing":
                call("mdm_index_mask", ptr(amount), 1, ptr(rng), stream_id, n, c, HW, ptr(mi), stream())
                call("mdm_degrade", ptr(x0_hat), None, ptr(mi), None, 1, ptr(rng), stream_id, n, c, HW, c, fm, fc, ptr(out), ptr(mask_out), ptr(mp), stream())
            else:
                call("mdm_degrade", ptr(x0_hat), None, None, ptr(amount), 1, ptr(rng), stream_id, n, c, HW, Cm, fm, fc, ptr(out), ptr(mask_out), ptr(mp), stream())

        def emit(update):
            call("mdm_sampler_step_params", ptr(ts_dev), T, ptr(ctr), ptr(ratio_tab), ptr(amount_tab), n, ptr(model.t_in),
                 ptr(ratio) if kind != 0 else None, ptr(amt_t), ptr(amt_next), ptr(rng), stream())
            call("mdm_shift", ptr(x_t), None, ptr(ratio) if kind != 0 else None, ptr(rng), 2, kind, float(getattr(a, "noise_mean", 0.0)),
                 per_col, n, c, hw, hw, ptr(s), ptr(x_in), model.dt, ptr(model.x_in.data), model.cin_p, stream())
            _lib._recording.extend(model.forward_plan)
            call("mdm_sampler_x0", model.dt, ptr(model.y_out.data), model.cout_p, ptr(x_in), ptr(s), n, c, hw, hw, None, None,
                 ptr(x0_hat), stream())
            if dep == "independent":
                degrade(amt_t, 3, None, d_t, m_t, mi_t)
                degrade(amt_next, 4, None, d_next, m_next, mi_next)
            elif dep == "dependent_t":                    # nested masks: the SAME Philox stream thresholded at both ratios
                degrade(amt_t, 3, None, d_t, m_t, mi_t)
                degrade(amt_next, 3, None, d_next, m_next, mi_next)
            else:
                degrade(None, 0, m_next, d_t, None, None)
                degrade(amt_next, 4, None, d_next, m_next, mi_next)
            if update:
                call("mdm_sampler_update", ptr(d_t), ptr(d_next), ptr(x_t), ptr(diff), int(mode == "base_momentum"), x_t.numel(), stream())

        with _lib.Recording() as body:
            emit(True)
        with _lib.Recording() as last:
            emit(False)                                      # i == 0: base_sampling breaks before the write, momentum skips it
        keep = (ts_dev, ctr, ratio, amt_t, amt_next, s, x_in, d_t, d_next, m_t, m_next, mi_t, mi_next, mp, diff, amount_tab, ratio_tab)
        gb, gl = _lib.GraphExec(body), _lib.GraphExec(last)
        for _ in range(T - 1):
            gb.launch()
        gl.launch()
        self._graph_keep = (keep, gb, gl)
        return x0_hat

    def _sample_mean_shift_momentum(self, model, timesteps, latent=None):
        from .unet import UNet
        a, S = self.args, self.Scheduler
        dev = S.device
        T = len(timesteps)
        n, c, hw = a.sample_num, a.out_channel, a.data_size
        dep, mode = a.sampling_mask_dependency, a.momentum_adaptive
        if dep not in ("independent", "dependent_prev", "dependent_t"):
            raise UnboundLocalError(f"sampling_mask_dependency={dep!r} is not an option upstream")
        if dep == "dependent_t":                # runs upstream for a sub-case only (scheduler.py:480-549); same errors here
            S._check_dependent_args(a, a.mean_option, a.mean_area)
            S._check_degrade_args(torch.empty(1, c, 1, 1))
        if mode not in ("base_sampling", "base_momentum"):
            raise UnboundLocalError(f"momentum_adaptive={mode!r} does not run upstream (D4)")
        hist_mode = getattr(a, "sample_history", True)
        fused = isinstance(model, UNet)
        if fused and getattr(a, "sampler_uniform_t", True):
            model = model.with_uniform_t()      # every reverse step passes ONE timestep for the whole batch (sampler.py:137-145)
        if fused:
            assert (model.N, model.H, model.W) == (n, hw, hw), "UNet plan was built for another batch/extent"
        x_t = (latent if latent is not None else self._get_latent_initial(model)).to(dev, torch.float32).contiguous()
        m_next = torch.zeros(n, c, hw, hw, device=dev)
        hist = None
        if hist_mode:
            hist = {k: torch.zeros(T + 1, n, c, hw, hw, device=dev) for k in HISTORY_NAMES}
        if (fused and not hist_mode and S.rng_mode == "device" and model.use_graph and getattr(a, "sampler_graph", True)
                and self.step_hook is None):
            return self._sample_graph(model, timesteps, x_t, dep, mode), []
        x0_hat = torch.empty_like(x_t)
        pred_nchw = torch.empty_like(x_t) if hist_mode else None
        shifted0 = torch.empty_like(x_t) if hist_mode else None
        diff = torch.zeros_like(x_t)
        numel = x_t.numel()
        for i in range(T - 1, -1, -1):
            slot = T - i
            S.dev_rng.advance()
            time = torch.full((n,), float(timesteps[i]), device=dev)
            if self.step_hook is not None:
                self.step_hook(i, slot, x_t)
            if hist_mode:
                hist["sample_t"][slot].copy_(x_t)
            nhwc = (model.dt, model.x_in.data, model.cin_p) if fused else None
            s, x_in = S.shift_and_perturb(time, x_t, want_nhwc=nhwc)                 # sampler.py:142-143
            pred, cp, pdt = self._predict(model, x_in, time, fused)                  # :145
            call("mdm_sampler_x0", pdt, ptr(pred), cp, ptr(x_in), ptr(s), n, c, hw, hw, ptr(pred_nchw), ptr(shifted0),
                 ptr(x0_hat), stream())                                              # :146-152
            next_t = time - 1 if i > 0 else time                                     # :167-170
            n_t = S.get_black_area_num_pixels_time(time)
            n_next = S.get_black_area_num_pixels_time(next_t)
            if dep == "independent":                                                 # :175-181
                d_t, m_t, _ = S.degrade_independent_base_sampling(n_t, x0_hat, mean_option=a.mean_option, mean_area=a.mean_area, _stream=3)
                d_next, m_next, _ = S.degrade_independent_base_sampling(n_next, x0_hat, mean_option=a.mean_option, mean_area=a.mean_area, _stream=4)
            elif dep == "dependent_t":                                               # :191-196
                d_t, m_t, _, d_next, m_next, _ = S.degrade_dependent_base_sampling(n_t, n_next, x0_hat, mean_option=a.mean_option, mean_area=a.mean_area)
            else:                                                                    # :184-188
                m_t = None
                d_t = S.degrade_with_mask(x0_hat, m_next, mean_option=a.mean_option, mean_area=a.mean_area)
                d_next, m_next, _ = S.degrade_independent_base_sampling(n_next, x0_hat, mean_option=a.mean_option, mean_area=a.mean_area, _stream=4)
            if hist_mode:
                hist["shift"][slot].copy_(s); hist["shifted"][slot].copy_(x_in); hist["mask"][slot].copy_(pred_nchw)
                hist["shifted_result"][slot].copy_(shifted0); hist["sample_0"][slot].copy_(x0_hat)
                if dep in ("independent", "dependent_t"):
                    hist["degraded_mask"][slot].copy_(m_t); hist["degraded_mask_next"][slot].copy_(m_next)
                else:
                    hist["degraded_mask"][slot].copy_(m_next)
            if mode == "base_sampling" and i == 0:                                   # :204-205 (break before the writes)
                break
            if i > 0 or mode == "base_sampling":                                     # :206-216
                call("mdm_sampler_update", ptr(d_t), ptr(d_next), ptr(x_t), ptr(diff), int(mode == "base_momentum"), numel, stream())
            if hist_mode:
                hist["degraded_next_t"][slot].copy_(d_next); hist["degraded_t"][slot].copy_(d_t); hist["difference"][slot].copy_(diff)
        out_hist = []
        if hist_mode:
            out_hist = [hist[k] if hist_mode == "device" else hist[k].cpu() for k in HISTORY_NAMES]
        return x0_hat, out_hist

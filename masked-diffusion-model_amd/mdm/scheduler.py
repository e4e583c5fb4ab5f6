"""Forward-process scheduler on the GPU -- same surface as the reference `Scheduler(args)`.

Mirrors reference code/scheduler.py: `update_ddpm_num_steps` (:27-65), schedule tables
(:103-142), `get_timesteps_epoch` (:173-192), `get_black_area_num_pixels_time` (:88-100),
`degrade_training` (:266-323), `degrade_independent_base_sampling` (:418-477),
`degrade_with_mask` (:572-598), `get_schedule_shift_time` (:612-732), `perturb_shift(_inverse)`
(:757-777), `get_weight_timesteps` (:780-794).  Options that do not run upstream at HEAD
 raise the same kind of error here.

Randomness (`args.rng_mode`, default "replay"):
  replay : draws come from torch's CPU generator in exactly the reference's order and shape
           (scheduler.py:282,288,294,620,658,675,694,703-707) and are shipped to the device, so a
           run seeded like the reference reproduces its masks/shifts bit for bit;
  device : Philox4x32-10 inside the kernels (no host round trip; what bench.py and the
           captured train-step graph use).  Same distributions, different stream.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream

SHIFT_KINDS = {"non_shift": 0, "1-d_constant": 1, "3-d_constant": 2, "noise_reduction": 3,
               "noise_with_perturbation": 4, "noise_std_reduction": 5}


class DeviceRng:
    """Philox key/offset pair living in device memory.  `advance()` bumps the offset ON the device, in stream order
    (recorded like any other launch, so a captured step bumps it on every replay); a host-side counter copied over
    asynchronously would be read late by a GPU that the host runs ahead of."""

    def __init__(self, device, seed=0, rank=0):
        # key = (seed, rank): data-parallel ranks and sampler shards draw from disjoint streams (the reference seeds
        # every rank alike, main_train_masked.py:441-445 -- all ranks would draw identical t / masks)
        key = (int(seed) & 0xFFFFFFFF) | ((int(rank) & 0x7FFFFFFF) << 32)
        self.dev = torch.tensor([key, 0], dtype=torch.int64, device=device) if torch.cuda.is_available() \
            else torch.tensor([key, 0], dtype=torch.int64)

    def advance(self):
        call("mdm_rng_advance", ptr(self.dev), stream())


def _fill_mode(mean_option, mean_area):
    """-> (fill_mode, fill_const) of mdm_degrade; float(mean_option) first, like scheduler.py:298-317."""
    try:
        return 0, float(mean_option)
    except ValueError:
        pass
    if mean_option == "degraded_area":
        if mean_area == "image-wise":
            return 1, 0.0
        if mean_area == "channel-wise":
            return 2, 0.0
    elif mean_option == "non_degraded_area":
        return 3, 0.0
    raise UnboundLocalError(f"mean_pixel undefined for mean_option={mean_option!r}, mean_area={mean_area!r}")


class Scheduler:
    def __init__(self, args, device=None):
        _lib.load()
        self.args = args
        self.height = self.width = args.data_size
        self.image_size = self.height * self.width
        self.updated_ddpm_num_steps = None
        self.ratio_list = None
        self.black_area_pixels = None
        if device is not None:
            self.device = torch.device(device)
        elif torch.cuda.is_available():
            self.device = torch.device("cuda", torch.cuda.current_device())
        else:       # host-only use: schedule tables, timestep lists, gathers (every kernel call needs the GPU)
            self.device = torch.device("cpu")
        self.rng_mode = getattr(args, "rng_mode", "replay")
        self.reference_quirks = getattr(args, "reference_quirks", True)
        rank = getattr(args, "rng_rank", None)
        if rank is None:
            import torch.distributed as dist
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        self.dev_rng = DeviceRng(self.device, getattr(args, "seed", 0), rank)
        # Sharded sampling (mdm.Sampler over a process group): (lo, hi, n) = this rank computes rows [lo, hi) of an n-sample
        # batch.  The reference seeds every rank alike (main_train_masked.py:441-445), so a rank that drew only ITS rows from the
        # host generator would draw the same numbers as every other rank -- W copies of the same samples.  Host draws are
        # therefore made for all n samples on every rank (same seed -> same numbers) and sliced: the gathered batch equals
        # what one process computes.
        self.replay_rows = None

    def _host_rows(self, n_local):
        """(n to draw on the host, slice of it that is ours)."""
        if self.replay_rows is None:
            return n_local, slice(None)
        lo, hi, n = self.replay_rows
        assert hi - lo == n_local, (self.replay_rows, n_local)
        return n, slice(lo, hi)

    # ---- schedule tables (host, once per run) -------------------------------------
    def update_ddpm_num_steps(self, max_time=None):
        a = self.args          # the argument is ignored upstream too (scheduler.py:40-49, D15)
        kind, T = a.ddpm_schedule, a.ddpm_num_steps
        if kind == "linear":                                   # :103-109
            table = np.linspace(1e-3, 1, T)
            self.ratio_list = torch.tensor(table)
            self.black_area_pixels = self.ratio_list
        elif kind == "exponential":                            # :130-142
            e = getattr(a, "ddpm_schedule_base", 10.0) ** np.linspace(0, 1, T)
            self.ratio_list = torch.tensor(e / e[-1])
            self.black_area_pixels = self.ratio_list
        elif kind == "log":                                    # :112-127, 54-56
            if T > self.image_size:
                raise ValueError("Desired to remove number of pixels is greater than the size of input image.")
            v = np.log(np.linspace(1, self.image_size, T))
            v = v - v.min() + 1
            v = v * (self.image_size / v.max())
            counts = np.array(sorted(set(np.asarray(v, dtype=int))))
            counts[-1] = self.image_size
            self.black_area_pixels = counts
            self.ratio_list = torch.tensor(counts / self.image_size)
        elif kind == "sigmoid":
            raise TypeError("ddpm_schedule='sigmoid' does not run upstream (torch.flip of an ndarray, scheduler.py:58-61)")
        else:
            raise ValueError("Invalid mask ratio scheduler")
        self.updated_ddpm_num_steps = len(self.ratio_list)
        self.reverse_ratio = torch.flip(self.ratio_list, dims=(0,))
        # the device tables keep their addresses across calls (captured graphs hold raw pointers to them)
        def put(name, host):
            old = getattr(self, name, None)
            if old is not None and old.shape == host.shape and old.dtype == host.dtype:
                old.copy_(host)
            else:
                setattr(self, name, host.to(self.device))
        put("ratio_dev", self.ratio_list)
        put("pixels_dev", torch.as_tensor(np.asarray(self.black_area_pixels)))
        return self.updated_ddpm_num_steps

    def get_black_area_num_pixels_all(self):
        return self.black_area_pixels

    def get_updated_ddpm_num_steps(self):
        return self.updated_ddpm_num_steps

    def get_ratio_list(self):
        return self.ratio_list

    def get_reverse_ratio_list(self):
        return self.reverse_ratio

    def get_timesteps_epoch(self, epoch, epoch_length):
        scale = self.args.scheduler_num_scale_timesteps
        section = math.ceil((epoch + 1) / (epoch_length / scale))
        expo = scale - section
        stride = 2 ** expo if expo >= 0 else 1
        used = [i for i in range(1, self.updated_ddpm_num_steps + 1) if i % stride == 0]
        used[-1] = self.updated_ddpm_num_steps
        return used

    def get_black_area_num_pixels_time(self, time):
        idx = (time - 1).int() if hasattr(time, "int") else torch.as_tensor(time - 1).int()
        sel = self.args.select_degrade_pixel
        if sel == "indexing":
            tab = self.pixels_dev
        elif sel == "thresholding":
            tab = self.ratio_dev
        else:
            raise UnboundLocalError("select_degrade_pixel must be 'indexing' or 'thresholding'")
        return torch.index_select(tab.to(idx.device), 0, idx)

    def get_weight_timesteps(self, timesteps, power_base=2.0):
        alpha = torch.linspace(start=1, end=0, steps=self.updated_ddpm_num_steps)
        power = torch.pow(power_base, alpha).to(timesteps.device)
        return power[timesteps]

    # ---- masks / degrade ---------------------------------------------------------------
    def _check_degrade_args(self, img):
        sel, ch = self.args.select_degrade_pixel, getattr(self.args, "degrade_channel", None)
        if sel == "thresholding":
            if ch == "1-channel":
                return 1
            if ch == "3-channel":
                if img.shape[1] != 3:
                    raise RuntimeError("degrade_channel='3-channel' is hard-wired to 3 channels (scheduler.py:294-296)")
                return 3
            raise UnboundLocalError("thresholding needs degrade_channel in {'1-channel','3-channel'} (D8)")
        if sel == "indexing":
            return 1
        raise UnboundLocalError("select_degrade_pixel")

    def _degrade(self, amount, img, mean_option, mean_area, rng_stream, mask_in=None, want_mask=True, u=None, u_out=None):
        """Core of the degrade_* methods -> (x_t, mask, mean_pixel[N,C]).  `u`: uniforms already on the device to threshold
        (replay mode, thresholding) instead of drawing; `u_out`: a list that receives the uniforms this call drew."""
        img = img.to(self.device, torch.float32).contiguous()
        N, C, H, W = img.shape
        HW = H * W
        fm, fc = _fill_mode(mean_option, mean_area)
        x_t = torch.empty_like(img)
        mask = torch.empty_like(img) if want_mask else None
        mp = torch.empty(N, C, device=self.device)
        amt = None
        Cm = 1
        if mask_in is None:
            Cm = self._check_degrade_args(img)
            sel = self.args.select_degrade_pixel
            if sel == "indexing":
                if amount.dtype.is_floating_point:
                    raise TypeError("indexing needs integer pixel counts (linear/exponential schedules give ratios, D7)")
                if self.rng_mode == "replay":       # N serial CPU randperms, like scheduler.py:281-282
                    nh, rows = self._host_rows(N)
                    counts = amount.cpu()
                    if nh != N:                     # sharded sampler: every sample of a step has the same count
                        counts = counts[:1].expand(nh)
                    m = torch.ones(nh, HW)
                    for i, num in enumerate(counts):
                        m[i, torch.randperm(HW)[:num]] = 0.0
                    mask_in = m[rows].reshape(N, 1, H, W).expand(N, C, H, W).contiguous().to(self.device)
                else:
                    mask_in = torch.empty_like(img)
                    cnt = amount.to(self.device, torch.float64).contiguous()
                    call("mdm_index_mask", ptr(cnt), 1, ptr(self.dev_rng.dev), rng_stream, N, C, HW, ptr(mask_in), stream())
            else:
                amt = amount.to(self.device, torch.float64).contiguous()
                if self.rng_mode == "replay" and u is None:
                    nh, rows = self._host_rows(N)
                    u = torch.empty(nh, Cm * HW, dtype=torch.float32).uniform_(0.0, 1.0)[rows].contiguous().to(self.device)
                if u_out is not None:
                    u_out.append(u)
        else:
            mask_in = mask_in.to(self.device, torch.float32).contiguous()
        call("mdm_degrade", ptr(img), ptr(u), ptr(mask_in), ptr(amt), 1, ptr(self.dev_rng.dev), rng_stream, N, C, HW,
             Cm if mask_in is None else C, fm, fc, ptr(x_t), ptr(mask), ptr(mp), stream())
        return x_t, mask, mp

    def degrade_training(self, black_area_num, img, mean_option=None, mean_area=None):
        x_t, masks, mp = self._degrade(black_area_num, img, mean_option, mean_area, rng_stream=1)
        mp4 = mp[:, :, None, None]
        degrade_mask = (1 - masks) * mp4 + masks                         # scheduler.py:320 (visualisation)
        mean_mask = torch.ones_like(x_t) * mp4                            # scheduler.py:321
        return x_t, masks, degrade_mask, mean_mask

    def degrade_independent_base_sampling(self, black_area_num_t, img, mean_option=None, mean_area=None, _stream=3):
        x_t, masks, mp = self._degrade(black_area_num_t, img, mean_option, mean_area, rng_stream=_stream)
        return x_t, masks, mp[:, :, None, None] * torch.ones_like(x_t)

    def degrade_with_mask(self, img, masks, mean_option, mean_area):
        x_t, _, _ = self._degrade(None, img, mean_option, mean_area, rng_stream=0, mask_in=masks, want_mask=False)
        return x_t

    @staticmethod
    def _check_dependent_args(args, mean_option, mean_area):
        """What degrade_dependent_base_sampling runs for upstream (scheduler.py:480-549): 'thresholding' only (the 'indexing'
        branch is `pass`, :490-491), and mean_option 'degraded_area' or the STRING "0" -- there is no float() attempt in that
        function, so the int 0, numbers and 'non_degraded_area' leave mean_pixel_t unbound (D5)."""
        if args.select_degrade_pixel != "thresholding":
            raise UnboundLocalError("masks_t undefined: degrade_dependent_base_sampling has no 'indexing' branch (D5)")
        if not ((mean_option == "degraded_area" and mean_area in ("image-wise", "channel-wise"))
                or (isinstance(mean_option, str) and mean_option == "0")):
            raise UnboundLocalError(f"mean_pixel_t undefined for mean_option={mean_option!r}, mean_area={mean_area!r} (D5)")

    def degrade_dependent_base_sampling(self, black_area_num_t, black_area_num_next_t, img, mean_option, mean_area, _stream=3):
        """scheduler.py:480-549: the masks of t and t-1 are NESTED -- one uniform draw per pixel thresholded at both
        ratios -> (degraded_t, mask_t, mean_mask_t, degraded_next, mask_next, mean_mask_next).  Two launches of the degrade
        kernel over the SAME uniforms: the host draw shipped once (replay), or the same Philox stream id (device)."""
        self._check_dependent_args(self.args, mean_option, mean_area)
        drawn = []
        x_t, m_t, mp_t = self._degrade(black_area_num_t, img, mean_option, mean_area, rng_stream=_stream, u_out=drawn)
        x_n, m_n, mp_n = self._degrade(black_area_num_next_t, img, mean_option, mean_area, rng_stream=_stream, u=drawn[0] if drawn else None)
        ones = torch.ones_like(x_t)
        return x_t, m_t, mp_t[:, :, None, None] * ones, x_n, m_n, mp_n[:, :, None, None] * ones

    # ---- shift -------------------------------------------------------------------------
    def _shift_draws(self, n, C, H, W, ratio_cpu):
        """Replay-mode host draws in the reference's order/shape -> z tensor for mdm_shift (or None)."""
        st, mean = self.args.shift_type, float(getattr(self.args, "noise_mean", 0.0))
        if st == "1-d_constant":
            return torch.empty(n).uniform_(-1.0, 1.0)
        if st == "3-d_constant":
            return torch.empty(n, 3, 1, 1).uniform_(-1.0, 1.0)
        if st == "noise_reduction":
            return torch.empty(n, 1, H, W).normal_(mean=mean, std=1)
        if st == "noise_std_reduction":
            z = torch.zeros(n, 3, H, W)
            for i in range(n):
                z[i] = torch.empty(1, 3, H, W).normal_(mean=mean, std=float(ratio_cpu[i]))
            return z
        if st == "noise_with_perturbation":
            torch.empty((n,) if n == 1 else (n, 1, 1, 1)).uniform_(-1.0, 1.0)      # consumed then discarded (D11)
            return torch.empty(n, 3, H, W).normal_(mean=mean, std=1)
        return None

    def shift_and_perturb(self, timesteps, x_t, want_nhwc=None):
        """Fused get_schedule_shift_time + perturb_shift: -> (s, x_in); optionally also writes x_in as
        NHWC into `want_nhwc = (dtype, tensor, Cp)` for the U-Net."""
        st = self.args.shift_type
        if st not in SHIFT_KINDS:
            raise UnboundLocalError(f"shift_time undefined for shift_type={st!r}")
        kind = SHIFT_KINDS[st]
        x_t = x_t.to(self.device, torch.float32).contiguous()
        N, C, H, W = x_t.shape
        ratio = z = None
        if kind != 0:
            idx = (timesteps.int() - 1).to(self.device)
            ratio = torch.index_select(self.ratio_dev, 0, idx).contiguous()
            if self.rng_mode == "replay":
                nh, rows = self._host_rows(N)
                rc = ratio.cpu()
                z = self._shift_draws(nh, C, H, W, rc if nh == N else rc[:1].expand(nh))[rows].to(self.device).contiguous()
        # reference broadcast quirk: [N,*,H,W] * [N] lines up with the LAST axis when W == N (D10)
        per_col = int(self.reference_quirks and kind in (3, 4) and N == W and N > 1)
        s = torch.empty_like(x_t)
        x_in = torch.empty_like(x_t)
        dt, nh, Cp = want_nhwc if want_nhwc is not None else (0, None, 0)
        call("mdm_shift", ptr(x_t), ptr(z), ptr(ratio), ptr(self.dev_rng.dev), 2, kind,
             float(getattr(self.args, "noise_mean", 0.0)), per_col, N, C, H, W, ptr(s), ptr(x_in), dt, ptr(nh), Cp, stream())
        return s, x_in

    def get_schedule_shift_time(self, timesteps, binarymasks):
        zero = torch.zeros(binarymasks.shape, device=self.device, dtype=torch.float32)
        s, _ = self.shift_and_perturb(timesteps, zero)
        return s.to(getattr(self.args, "weight_dtype", torch.float32))

    def perturb_shift(self, data, shift):
        return data + shift.to(data.device)

    def perturb_shift_inverse(self, data, shift):
        return data - shift.to(data.device)

"""CPU oracle: forward-process scheduler (TEST INFRASTRUCTURE ONLY).

Own-words restatement of /root/reference/code/scheduler.py.  All randomness is
drawn through an `rng` object so a test can (a) let it consume torch's global
CPU generator in exactly the reference's order (bit-identical to the reference
under the same seed) or (b) record / replay the draws and hand the very same
numbers to the HIP path.
"""
from __future__ import annotations

import math

import numpy as np
import torch


# --------------------------------------------------------------------------- #
# RNG plumbing (reference draws: scheduler.py:282,288,294,440,446,620,658,675,
# 694,703,705,707 -- legacy torch.FloatTensor(...).uniform_/normal_ on the CPU
# default generator, and torch.randperm)
# --------------------------------------------------------------------------- #
class TorchGlobalRng:
    """Draws from torch's default CPU generator, like the reference does."""

    def uniform(self, shape, lo, hi):
        return torch.empty(tuple(shape), dtype=torch.float32).uniform_(lo, hi)

    def normal(self, shape, mean, std):
        return torch.empty(tuple(shape), dtype=torch.float32).normal_(mean=mean, std=std)

    def randperm(self, n):
        return torch.randperm(n)

    def randn(self, shape):
        return torch.randn(tuple(shape))

    def rand(self, shape):
        return torch.rand(tuple(shape))

    def randint(self, lo, hi, shape):
        return torch.randint(low=lo, high=hi, size=tuple(shape))


class RecordingRng(TorchGlobalRng):
    """TorchGlobalRng that also logs (kind, tensor) for every draw."""

    def __init__(self):
        self.log = []

    def _rec(self, kind, t):
        self.log.append((kind, t.clone()))
        return t

    def uniform(self, shape, lo, hi):
        return self._rec("uniform", super().uniform(shape, lo, hi))

    def normal(self, shape, mean, std):
        return self._rec("normal", super().normal(shape, mean, std))

    def randperm(self, n):
        return self._rec("randperm", super().randperm(n))

    def randn(self, shape):
        return self._rec("randn", super().randn(shape))

    def rand(self, shape):
        return self._rec("rand", super().rand(shape))

    def randint(self, lo, hi, shape):
        return self._rec("randint", super().randint(lo, hi, shape))


class ReplayRng:
    """Hands back a pre-recorded list of draws, checking kind and shape."""

    def __init__(self, log):
        self.log = list(log)
        self.pos = 0

    def _next(self, kind, shape=None):
        k, t = self.log[self.pos]
        self.pos += 1
        assert k == kind, f"replay: wanted {kind}, log has {k}"
        if shape is not None:
            assert tuple(t.shape) == tuple(shape), (kind, tuple(t.shape), tuple(shape))
        return t.clone()

    def uniform(self, shape, lo, hi):
        return self._next("uniform", shape)

    def normal(self, shape, mean, std):
        return self._next("normal", shape)

    def randperm(self, n):
        return self._next("randperm", (n,))

    def randn(self, shape):
        return self._next("randn", shape)

    def rand(self, shape):
        return self._next("rand", shape)

    def randint(self, lo, hi, shape):
        return self._next("randint", shape)


# --------------------------------------------------------------------------- #
# schedule tables
# --------------------------------------------------------------------------- #
def schedule_table(kind: str, num_steps: int, image_size: int, base: float = 10.0):
    """(ratio_list f64 tensor, black_area_pixels, T').

    scheduler.py:27-65 with the three live generators :103-109 (linear),
    :112-127 (log), :130-142 (exponential).  `sigmoid` is dead upstream (D6).
    """
    if kind == "linear":
        ratios = torch.tensor(np.linspace(1e-3, 1, num_steps))
        return ratios, ratios, num_steps
    if kind == "exponential":
        e = base ** np.linspace(0, 1, num_steps)
        ratios = torch.tensor(e / e[-1])
        return ratios, ratios, num_steps
    if kind == "log":
        if num_steps > image_size:
            raise ValueError("Desired to remove number of pixels is greater than the size of input image.")
        v = np.log(np.linspace(1, image_size, num_steps))
        v = v - v.min() + 1
        v = v * (image_size / v.max())
        counts = np.array(sorted(set(np.asarray(v, dtype=int))))
        counts[-1] = image_size                     # last step blanks every pixel (:55)
        ratios = torch.tensor(counts / image_size)  # (:56)
        return ratios, counts, len(counts)
    raise ValueError("Invalid mask ratio scheduler")


def timesteps_epoch(updated_steps: int, scale: int, epoch: int, epoch_length: int):
    """1-based timestep subset for one epoch, last forced to T' (scheduler.py:173-192)."""
    section = math.ceil((epoch + 1) / (epoch_length / scale))
    expo = scale - section
    stride = 2 ** expo if expo >= 0 else 1   # np.power(2, negative int) raises -> fallback (:187-188)
    used = [i for i in range(1, updated_steps + 1) if i % stride == 0]
    used[-1] = updated_steps
    return used


def table_at(table, t: torch.Tensor):
    """table[t-1] (scheduler.py:88-100); t may be float or int tensor."""
    idx = (t - 1).int()
    return torch.index_select(torch.as_tensor(table), 0, idx)


def loss_weights(updated_steps: int, timeindex: torch.Tensor, power_base: float):
    """power_base ** linspace(1,0,T')[timeindex] (scheduler.py:780-794)."""
    alpha = torch.linspace(start=1, end=0, steps=updated_steps)
    return torch.pow(power_base, alpha)[timeindex]


# --------------------------------------------------------------------------- #
# masks, fill value, degrade
# --------------------------------------------------------------------------- #
def draw_mask(rng, select: str, degrade_channel, amount: torch.Tensor, img_shape, height, width):
    """Binary keep-mask [N,C,H,W] (1 = pixel kept, 0 = degraded).

    scheduler.py:278-296 (training) == :430-448 (sampling).  `amount` is the
    per-sample count (indexing) or ratio (thresholding).
    """
    n, c = img_shape[0], img_shape[1]
    hw = height * width
    if select == "indexing":
        m = torch.ones(len(amount), hw)
        for i, num in enumerate(amount):
            m[i, rng.randperm(hw)[:num]] = 0.0
        return m.reshape(len(amount), 1, height, width).expand(n, c, height, width)
    if select == "thresholding":
        if degrade_channel == "1-channel":
            u = rng.uniform((n, hw), 0.0, 1.0)
            m = (u > amount.unsqueeze(1)).float()
            return m.reshape(len(amount), 1, height, width).expand(n, c, height, width)
        if degrade_channel == "3-channel":
            u = rng.uniform((n, 3 * hw), 0.0, 1.0)
            m = (u > amount.unsqueeze(1)).float()
            return m.reshape(len(amount), 3, height, width)
    raise UnboundLocalError("masks undefined for this select/degrade_channel combination (D8)")


def fill_value(img, masks, mean_option, mean_area):
    """Per-sample (/channel) fill for the degraded pixels, shape [N,C|1,1,1].

    scheduler.py:298-317 / :450-471 / :573-594.
    """
    n, c = img.shape[0], img.shape[1]
    try:
        return torch.ones(n, c, 1, 1) * float(mean_option)
    except ValueError:
        pass
    gone = 1 - masks
    if mean_option == "degraded_area":
        dims = (1, 2, 3) if mean_area == "image-wise" else (2, 3)
        return (img * gone).sum(dim=dims, keepdim=True) / gone.sum(dim=dims, keepdim=True)
    if mean_option == "non_degraded_area":
        mp = (img * masks).sum(dim=(2, 3), keepdim=True) / gone.sum(dim=(2, 3), keepdim=True) * -1
        mp[torch.isnan(mp)] = 0.0
        return mp
    raise UnboundLocalError("mean_pixel undefined for mean_option=%r" % (mean_option,))


def apply_degrade(img, masks, mean_pixel):
    """x_t = (1-m)*fill + m*x0 (scheduler.py:319, :473, :596)."""
    return (1 - masks) * mean_pixel + masks * img


# --------------------------------------------------------------------------- #
# shift
# --------------------------------------------------------------------------- #
def _times_ratio(rnd, ratio):
    """`random * ratio` with the reference's try/except broadcast (D10).

    scheduler.py:679-684, 712-717: a [N,*,H,W] tensor times a [N] vector
    broadcasts along the LAST axis whenever W == N (or N == 1); only otherwise
    does the RuntimeError fallback scale per sample.
    """
    if rnd.shape[-1] == ratio.shape[0] or ratio.shape[0] == 1 or rnd.shape[-1] == 1:
        return rnd * ratio
    return rnd * ratio[:, None, None, None].expand_as(rnd)


def shift_time(rng, shift_type, timesteps, ratio_list, mask_like, height, width,
               noise_mean=0.0, weight_dtype=torch.float32):
    """Shift tensor expanded to mask_like's shape (scheduler.py:612-732)."""
    t = timesteps.int()
    n = len(t)
    if shift_type == "1-d_constant":
        r = rng.uniform((n,), -1.0, 1.0)
        ratio = torch.index_select(ratio_list, 0, t - 1)
        s = (r * ratio).to(weight_dtype)[:, None, None, None]
    elif shift_type == "3-d_constant":
        r = torch.ones(n, 3, 1, 1) * rng.uniform((n, 3, 1, 1), -1.0, 1.0)
        ratio = torch.index_select(ratio_list, 0, t - 1)[:, None, None, None].expand_as(r)
        s = (r * ratio).to(weight_dtype)
    elif shift_type == "noise_reduction":
        z = rng.normal((n, 1, height, width), noise_mean, 1)
        ratio = torch.index_select(ratio_list, 0, t - 1)
        s = _times_ratio(z, ratio)
    elif shift_type == "noise_std_reduction":
        ratio = torch.index_select(ratio_list, 0, t - 1)
        s = torch.zeros(n, 3, height, width)
        for i in range(n):
            s[i] = rng.normal((1, 3, height, width), noise_mean, float(1 * ratio[i]))
    elif shift_type == "noise_with_perturbation":
        # the uniform draw is consumed, then discarded (D11, :700-717)
        rng.uniform((n,) if n == 1 else (n, 1, 1, 1), -1.0, 1.0)
        z = rng.normal((n, 3, height, width), noise_mean, 1)
        ratio = torch.index_select(ratio_list, 0, t - 1)
        s = _times_ratio(z, ratio)
    elif shift_type == "non_shift":
        s = torch.zeros(n, 3, height, width)
    else:
        raise UnboundLocalError("shift_time undefined for shift_type=%r" % (shift_type,))
    return s.to(weight_dtype).expand_as(mask_like)


# --------------------------------------------------------------------------- #
# class with the reference's method names
# --------------------------------------------------------------------------- #
class SchedulerRef:
    """Same surface as reference `Scheduler(args)` (scheduler.py:13-794)."""

    def __init__(self, args, rng=None):
        self.args = args
        self.height = self.width = args.data_size
        self.image_size = self.height * self.width
        self.updated_ddpm_num_steps = None
        self.ratio_list = None
        self.black_area_pixels = None
        self.rng = rng or TorchGlobalRng()

    def update_ddpm_num_steps(self, max_time=None):
        # the argument is ignored upstream too (D15)
        r, px, steps = schedule_table(self.args.ddpm_schedule, self.args.ddpm_num_steps,
                                      self.image_size, getattr(self.args, "ddpm_schedule_base", 10.0))
        self.ratio_list, self.black_area_pixels, self.updated_ddpm_num_steps = r, px, steps
        self.reverse_ratio = torch.flip(r, dims=(0,))
        return steps

    def get_black_area_num_pixels_all(self):
        return self.black_area_pixels

    def get_updated_ddpm_num_steps(self):
        return self.updated_ddpm_num_steps

    def get_ratio_list(self):
        return self.ratio_list

    def get_reverse_ratio_list(self):
        return self.reverse_ratio

    def get_black_area_num_pixels_time(self, time):
        if self.args.select_degrade_pixel == "indexing":
            return table_at(self.black_area_pixels, time)
        if self.args.select_degrade_pixel == "thresholding":
            return table_at(self.ratio_list, time)
        raise UnboundLocalError("select_degrade_pixel")

    def get_timesteps_epoch(self, epoch, epoch_length):
        return timesteps_epoch(self.updated_ddpm_num_steps, self.args.scheduler_num_scale_timesteps,
                               epoch, epoch_length)

    def _mask(self, amount, img):
        return draw_mask(self.rng, self.args.select_degrade_pixel, getattr(self.args, "degrade_channel", None),
                         amount, img.shape, self.height, self.width)

    def degrade_training(self, black_area_num, img, mean_option=None, mean_area=None):
        masks = self._mask(black_area_num, img)
        mp = fill_value(img, masks, mean_option, mean_area)
        x_t = apply_degrade(img, masks, mp)
        degrade_mask = (1 - masks) * mp + masks                              # :320
        mean_mask = torch.ones(len(black_area_num), img.shape[1], self.height, self.width) * mp   # :321
        return x_t, masks, degrade_mask, mean_mask

    def degrade_independent_base_sampling(self, black_area_num_t, img, mean_option=None, mean_area=None):
        masks = self._mask(black_area_num_t, img)
        mp = fill_value(img, masks, mean_option, mean_area)
        x_t = apply_degrade(img, masks, mp)
        mean_mask = mp * torch.ones(len(black_area_num_t), masks.shape[1], self.height, self.width)
        return x_t, masks, mean_mask

    def degrade_with_mask(self, img, masks, mean_option, mean_area):
        return apply_degrade(img, masks, fill_value(img, masks, mean_option, mean_area))

    def degrade_dependent_base_sampling(self, black_area_num_t, black_area_num_next_t, img, mean_option, mean_area):
        """NESTED masks for t and t-1 from ONE uniform draw thresholded twice (scheduler.py:480-549).  Runs upstream only
        for 'thresholding' (no 'indexing' branch, :490-491) and for mean_option 'degraded_area' or the STRING "0" (:514-535:
        there is no float() attempt here, so the int 0 and every other value leave mean_pixel unbound)."""
        a = self.args
        n, c, hw = img.shape[0], img.shape[1], self.height * self.width
        if a.select_degrade_pixel != "thresholding":
            raise UnboundLocalError("masks_t undefined: degrade_dependent_base_sampling has no 'indexing' branch (D5)")
        ch = getattr(a, "degrade_channel", None)
        if ch == "1-channel":
            u = self.rng.uniform((n, hw), 0.0, 1.0)
            shape = lambda m: m.reshape(n, 1, self.height, self.width).expand(n, c, self.height, self.width)
        elif ch == "3-channel":
            u = self.rng.uniform((n, 3 * hw), 0.0, 1.0)
            shape = lambda m: m.reshape(n, 3, self.height, self.width)
        else:
            raise UnboundLocalError("masks_t undefined for this degrade_channel (D8)")
        m_t = shape((u > black_area_num_t.unsqueeze(1)).float())
        m_next = shape((u > black_area_num_next_t.unsqueeze(1)).float())
        if mean_option == "degraded_area" and mean_area in ("image-wise", "channel-wise"):
            mp_t, mp_next = fill_value(img, m_t, mean_option, mean_area), fill_value(img, m_next, mean_option, mean_area)
        elif isinstance(mean_option, str) and mean_option == "0":
            mp_t = mp_next = torch.ones(n, c, 1, 1) * 0.0
        else:
            raise UnboundLocalError("mean_pixel_t undefined for mean_option=%r (D5)" % (mean_option,))
        ones = torch.ones(n, m_t.shape[1], self.height, self.width)
        return (apply_degrade(img, m_t, mp_t), m_t, mp_t * ones, apply_degrade(img, m_next, mp_next), m_next, mp_next * ones)

    def get_schedule_shift_time(self, timesteps, binarymasks):
        return shift_time(self.rng, self.args.shift_type, timesteps, self.ratio_list, binarymasks,
                          self.height, self.width, getattr(self.args, "noise_mean", 0.0),
                          getattr(self.args, "weight_dtype", torch.float32))

    @staticmethod
    def perturb_shift(data, shift):
        return data + shift          # :757-766 (the except branch is only for 1-D shifts)

    @staticmethod
    def perturb_shift_inverse(data, shift):
        return data - shift          # :769-777

    def get_weight_timesteps(self, timesteps, power_base=2.0):
        return loss_weights(self.updated_ddpm_num_steps, timesteps, power_base)

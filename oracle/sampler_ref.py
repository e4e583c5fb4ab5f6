"""CPU oracle: reverse sampler (TEST INFRASTRUCTURE ONLY).

Own-words restatement of /root/reference/code/sampler.py:46-83 (initial
latent) and :109-261 (`_sample_mean_shift_momentum`), for the option values
that run at HEAD: sampling_mask_dependency in {independent,
dependent_prev, dependent_t (thresholding with mean_option 'degraded_area' or "0")},
momentum_adaptive in {base_sampling, base_momentum}.
"""
from __future__ import annotations

import numpy as np
import torch

from .scheduler_ref import TorchGlobalRng

HISTORY_NAMES = ["sample_t", "shift", "shifted", "mask", "shifted_result", "sample_0",
                 "degraded_mask", "degraded_mask_next", "degraded_t", "difference", "degraded_next_t"]


class SamplerRef:
    def __init__(self, dataset, args, Scheduler, dataset_hist, rng=None):
        self.dataset, self.args, self.Scheduler, self.dataset_hist = dataset, args, Scheduler, dataset_hist
        self.rng = rng or getattr(Scheduler, "rng", None) or TorchGlobalRng()

    def _get_latent_initial(self, model=None):
        """Constant-colour start image per sample (sampler.py:46-83)."""
        a = self.args
        d = 1 if a.mean_area == "image-wise" else 3
        shape = a.sample_latent_shape.lower()
        if shape == "data":
            hshape, edges, cum = self.dataset_hist
            idx = torch.searchsorted(cum, self.rng.rand((a.sample_num,)))
            idx = np.unravel_index(idx, hshape)
            cols = []
            for c in range(d):
                r = self.rng.rand((a.sample_num,))
                lo, hi = edges[c][idx[c]], edges[c][idx[c] + 1]
                cols.append(((hi - lo) * r + lo).unsqueeze(-1))
            mean = torch.cat([torch.empty(a.sample_num, 0)] + cols, 1)
        elif shape == "zero":
            mean = torch.zeros(a.sample_num, d)
        elif shape == "normal":
            mean = self.rng.randn((a.sample_num, d))
        elif shape == "uniform":
            mean = self.rng.uniform((a.sample_num, d), -1, 1)
        else:
            raise IndexError("sample_latent_shape=%r is broken upstream" % shape)
        return mean[:, :, None, None].expand(a.sample_num, a.out_channel, a.data_size, a.data_size)

    def sample(self, model, timesteps_used_epoch, interpolation_shift=None):
        return self._sample_mean_shift_momentum(model, timesteps_used_epoch)

    def _sample_mean_shift_momentum(self, model, timesteps):
        a, S = self.args, self.Scheduler
        T = len(timesteps)
        n, c, hw = a.sample_num, a.out_channel, a.data_size
        x_t = self._get_latent_initial(model).clone()
        m_t = torch.zeros(n, c, hw, hw)
        m_next = torch.zeros(n, c, hw, hw)
        hist = {k: torch.zeros(T + 1, n, c, hw, hw) for k in HISTORY_NAMES}
        x0_hat = None
        with torch.no_grad():
            for i in range(T - 1, -1, -1):
                slot = T - i
                time = torch.Tensor([timesteps[i]]).expand(n)
                s = S.get_schedule_shift_time(time, m_t)                    # :142
                x_in = S.perturb_shift(x_t, s)                              # :143
                pred = model(x_in, time).sample                             # :145
                shifted0 = x_in + pred                                      # :146
                x0_hat = S.perturb_shift_inverse(shifted0, s)               # :152
                hist["sample_t"][slot] = x_t; hist["shift"][slot] = s; hist["shifted"][slot] = x_in
                hist["mask"][slot] = pred; hist["shifted_result"][slot] = shifted0; hist["sample_0"][slot] = x0_hat
                next_t = time - 1 if i > 0 else time                        # :167-170
                n_t = S.get_black_area_num_pixels_time(time)
                n_next = S.get_black_area_num_pixels_time(next_t)
                dep = a.sampling_mask_dependency
                if dep == "independent":                                    # :175-181
                    d_t, m_t, _ = S.degrade_independent_base_sampling(n_t, x0_hat, mean_option=a.mean_option, mean_area=a.mean_area)
                    d_next, m_next, _ = S.degrade_independent_base_sampling(n_next, x0_hat, mean_option=a.mean_option, mean_area=a.mean_area)
                    hist["degraded_mask"][slot] = m_t; hist["degraded_mask_next"][slot] = m_next
                elif dep == "dependent_prev":                               # :184-188
                    d_t = S.degrade_with_mask(x0_hat, m_next, mean_option=a.mean_option, mean_area=a.mean_area)
                    d_next, m_next, _ = S.degrade_independent_base_sampling(n_next, x0_hat, mean_option=a.mean_option, mean_area=a.mean_area)
                    hist["degraded_mask"][slot] = m_next
                elif dep == "dependent_t":                                  # :191-196
                    d_t, m_t, _, d_next, m_next, _ = S.degrade_dependent_base_sampling(n_t, n_next, x0_hat, mean_option=a.mean_option, mean_area=a.mean_area)
                    hist["degraded_mask"][slot] = m_t; hist["degraded_mask_next"][slot] = m_next
                else:
                    raise UnboundLocalError("sampling_mask_dependency=%r is not an option upstream" % dep)
                mode = a.momentum_adaptive
                if mode == "base_sampling":                                 # :199-207
                    if i == 0:
                        break
                    diff = d_next - d_t
                    x_t = d_next
                elif mode == "base_momentum":                               # :209-216
                    if i > 0:
                        diff = d_next - d_t
                        x_t = x_t + diff
                else:
                    raise UnboundLocalError("momentum_adaptive=%r does not run upstream (D4)" % mode)
                hist["degraded_next_t"][slot] = d_next; hist["degraded_t"][slot] = d_t; hist["difference"][slot] = diff
        return x0_hat, [hist[k] for k in HISTORY_NAMES]

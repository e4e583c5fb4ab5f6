"""CPU oracle: one masked-diffusion train step (TEST INFRASTRUCTURE ONLY).

Own-words restatement of trainer_masked_mean_shift.py:82-193 (mean-shift
trainer) and trainer_masked.py:95-183 (base trainer == mean-shift with
`shift_type=non_shift`), without accelerate: backward, global-norm
clip at 1.0, optimizer step, EMA.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def ema_decay(step, max_decay=0.9999, inv_gamma=1.0, power=0.75, min_decay=0.0, use_warmup=True):
    """diffusers.EMAModel.get_decay semantics per the call-site arguments
    (main_train_masked.py:119-127).  Third-party: NOT oracle-checked (parity unpinned)."""
    step = max(0, step - 1)
    if step <= 0:
        return 0.0
    v = 1 - (1 + step / inv_gamma) ** -power if use_warmup else (1 + step) / (10 + step)
    return max(min(v, max_decay), min_decay)


def train_step_ref(model, opt, sched, args, x0, timesteps_used_epoch, rng, ema_params=None, ema_step=0,
                   mean_shift=True, do_update=True):
    """Returns dict(loss, pred, x_t, mask, shift, x_in, recon, timeindex, t, grad_norm).

    Order of RNG draws: randint(timeindex) -> mask draw -> shift draws.
    """
    n = x0.shape[0]
    x0 = x0.to(args.weight_dtype)
    timeindex = rng.randint(0, len(timesteps_used_epoch), (n,))                       # ms:109 / base:114
    t = torch.index_select(torch.tensor(timesteps_used_epoch), 0, timeindex)
    if mean_shift:
        t = t.to(args.weight_dtype)                                                   # ms:110 (D12)
    amount = sched.get_black_area_num_pixels_time(t)                                  # ms:112
    x_t, masks, _, _ = sched.degrade_training(amount, x0, mean_option=args.mean_option, mean_area=args.mean_area)
    if mean_shift:
        s = sched.get_schedule_shift_time(t, masks).to(args.weight_dtype)             # ms:119
        x_in = sched.perturb_shift(x_t.to(args.weight_dtype), s)                      # ms:120
    else:
        s = torch.zeros_like(x_t)
        x_in = x_t
    pred = model(x_in, t).sample                                                      # ms:140
    recon = x_in + pred                                                               # ms:142
    inv = sched.perturb_shift_inverse(recon, s) if mean_shift else recon              # ms:145
    w = sched.get_weight_timesteps(timeindex, args.loss_weight_power_base) if getattr(args, "loss_weight_use", False) else None
    if mean_shift:
        per = F.mse_loss(inv.float(), x0.float(), reduction="none")                   # ms:153
    else:
        per = F.mse_loss(inv, x0, reduction="none")                                   # base:134
    if w is not None:
        per = w[:, None, None, None] * per
    loss = per.mean()
    out = dict(loss=loss.detach(), pred=pred.detach(), x_t=x_t, mask=masks, shift=s, x_in=x_in,
               recon=inv.detach(), timeindex=timeindex, t=t)
    if do_update:
        opt.zero_grad()
        loss.backward()
        out["grad_norm"] = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)    # ms:163-164
        opt.step()
        if ema_params is not None:                                                    # ms:170-172
            d = ema_decay(ema_step + 1, args.ema_max_decay, args.ema_inv_gamma, args.ema_power)
            with torch.no_grad():
                for e, p in zip(ema_params, model.parameters()):
                    e.sub_((1 - d) * (e - p))
            out["ema_decay"] = d
    return out

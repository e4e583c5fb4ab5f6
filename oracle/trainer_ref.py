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
                   mean_shift=True, do_update=True, grad_scale=1.0, sync=True, zero=True):
    """Returns dict(loss, pred, x_t, mask, shift, x_in, recon, timeindex, t, grad_norm).

    Order of RNG draws: randint(timeindex) -> mask draw -> shift draws.
    Gradient accumulation (`accelerator.accumulate`, ms:139-172): `grad_scale` = 1 / gradient_accumulation_steps is what
    `accelerator.backward` multiplies the loss by, `zero` = this is the first micro-step after an update (the prepared
    optimizer's zero_grad is skipped on non-syncing steps), `sync` = clip + optimizer step + EMA happen on this one.
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
        if zero:
            opt.zero_grad()
        (loss * grad_scale if grad_scale != 1.0 else loss).backward()
        if not sync:
            return out
        out["grad_norm"] = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)    # ms:163-164
        opt.step()
        if ema_params is not None:                                                    # ms:170-172
            d = ema_decay(ema_step + 1, args.ema_max_decay, args.ema_inv_gamma, args.ema_power)
            with torch.no_grad():
                for e, p in zip(ema_params, model.parameters()):
                    e.sub_((1 - d) * (e - p))
            out["ema_decay"] = d
    return out


def train_loop_ref(model, opt, lr_sched, sched, args, dataloader, epoch_start, epoch_length, grad_accum=1, num_processes=1,
                   split_batches=False, mean_shift=True):
    """`Trainer.train()` / `_run_epoch` (trainer_masked_mean_shift.py:196-273) as a plain loop, with what accelerate's
    prepared objects do around it (accelerate.Accelerator._do_sync, AcceleratedOptimizer, AcceleratedScheduler): a micro-step
    syncs when its count reaches `grad_accum` or it is the last batch of the dataloader; the optimizer steps and is cleared
    only then; the LR schedule steps only then, `num_processes` times unless `split_batches`.
    -> dict(losses, sync, lr, used (per epoch), global_step)."""
    args.updated_ddpm_num_steps = sched.update_ddpm_num_steps(args.ddpm_num_steps)
    losses, syncs, lrs, used_all = [], [], [], []
    global_step, micro = 0, 0
    fresh = True
    for epoch in range(epoch_start, epoch_start + epoch_length):
        used = sched.get_timesteps_epoch(epoch, epoch_length)
        used_all.append(list(used))
        n = len(dataloader)
        for i, batch in enumerate(dataloader):
            if i == n - 1:
                micro, sync = 0, True
            else:
                micro += 1
                sync = micro % grad_accum == 0
            r = train_step_ref(model, opt, sched, args, batch[0], used, sched.rng, mean_shift=mean_shift,
                               grad_scale=1.0 / grad_accum, sync=sync, zero=fresh)
            fresh = sync
            if sync:
                for _ in range(1 if split_batches else num_processes):
                    lr_sched.step()
                global_step += 1
            losses.append(float(r["loss"])); syncs.append(sync); lrs.append(lr_sched.get_last_lr()[0])
    return dict(losses=losses, sync=syncs, lr=lrs, used=used_all, global_step=global_step)

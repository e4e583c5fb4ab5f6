"""Optimizer-side objects the reference's trainers are handed: AdamW, EMA,
LR schedule, and a thin accelerator -- over the flat parameter buffers of mdm.UNet.

Reference call sites: optimizer `optim.AdamW(model.parameters(), lr=lr)` (main_train_masked.py:134-141,
torch defaults betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2); EMA `EMAModel(decay=ema_max_decay,
use_ema_warmup=True, inv_gamma, power)` (:116-131) stepped at trainer_masked_mean_shift.py:170-172;
LR schedules diffusers `get_*_schedule_with_warmup` (:144-165).  EMAModel and the LR schedules are
third-party code that is absent from the reference tree: their semantics are implemented from the
call-site arguments and are NOT oracle-checked ("parity unpinned").
"""
from __future__ import annotations

import math

import torch

from . import _lib, ops
from ._lib import call, ptr, stream


class AdamW:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        st = model.store
        self.model, self.store = model, st
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, initial_lr=lr)]
        self.m = torch.zeros_like(st.P)
        self.v = torch.zeros_like(st.P)
        self.t = 0
        # the 8-float hyper-parameter block travels by async H2D copy from pinned memory, which is read when the copy
        # EXECUTES: a host that runs ahead of the GPU must not overwrite a block whose copy has not run yet -> a ring of
        # pinned slots in groups of 16, each group guarded by one event recorded behind its last copy
        self._hp_slots = 64
        self.hp_host = torch.zeros(self._hp_slots, 8).pin_memory() if torch.cuda.is_available() else torch.zeros(self._hp_slots, 8)
        self._hp_events = [None] * (self._hp_slots // 16)
        self._hp_k = 0
        self.hp = torch.zeros(8, device=st.P.device)
        self.sqnorm = torch.zeros(1, device=st.P.device)

    def hyper(self, ema_decay=0.0, advance=True):
        """Refresh the 8-float device block read by mdm_adamw_ema (async H2D)."""
        g = self.param_groups[0]
        if advance:
            self.t += 1
        b1, b2 = g["betas"]
        k = self._hp_k
        self._hp_k = (k + 1) % self._hp_slots
        g16 = k // 16
        if k % 16 == 0 and self._hp_events[g16] is not None:
            self._hp_events[g16].synchronize()        # every copy that read this group of slots has executed
        h = self.hp_host[k]
        h[0] = g["lr"]; h[1] = b1; h[2] = b2; h[3] = g["eps"]
        h[4] = g["weight_decay"]
        h[5] = 1 - b1 ** self.t; h[6] = 1 - b2 ** self.t; h[7] = ema_decay
        self.hp.copy_(h, non_blocking=True)
        if self.hp.is_cuda and k % 16 == 15:
            ev = self._hp_events[g16] or torch.cuda.Event()
            ev.record()
            self._hp_events[g16] = ev

    def emit_update(self, ema_buf=None, max_norm=1.0, gmul=1.0):
        """Enqueue (or record) grad-norm + clip + AdamW + EMA + bf16 shadow over the flat buffers."""
        st = self.store
        call("mdm_sqnorm", ptr(st.G), st.size, ptr(self.sqnorm), stream())
        call("mdm_adamw_ema", ptr(st.P), ptr(st.G), ptr(self.m), ptr(self.v), ptr(ema_buf), ptr(st.Pb), st.size,
             ptr(self.hp), ptr(self.sqnorm), float(max_norm), float(gmul), stream())
        st.emit_transposed_shadow()
        st.emit_split_shadow()          # fp32 stores with split products: the filters' hi / lo shadow follows the weights

    def step(self, max_norm=0.0):
        self.hyper()
        self.emit_update(None, max_norm)

    def zero_grad(self):
        self.store.G.zero_()

    def grad_norm(self):
        return float(self.sqnorm.sqrt())

    def state_dict(self):
        """`torch.optim.AdamW.state_dict()` layout: per-parameter `step` / `exp_avg` / `exp_avg_sq` in the reference's
        shapes, indexed in `model.parameters()` order (what accelerate writes to optimizer.bin)."""
        order = self.model.reference_param_order()
        m = self.store.state_dict(order=order, src=self.m)
        v = self.store.state_dict(order=order, src=self.v)
        state = {i: dict(step=torch.tensor(float(self.t)), exp_avg=m[k], exp_avg_sq=v[k]) for i, k in enumerate(order)} if self.t else {}
        g = self.param_groups[0]
        group = dict(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"], amsgrad=False,
                     maximize=False, foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=True,
                     initial_lr=g["initial_lr"], params=list(range(len(order))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        order = self.model.reference_param_order()
        st = sd["state"]
        if st:
            assert len(st) == len(order), (len(st), len(order))
            dev = self.m.device
            self.m.copy_(self.store.flat_from_reference({k: st[i]["exp_avg"] for i, k in enumerate(order)}).to(dev))
            self.v.copy_(self.store.flat_from_reference({k: st[i]["exp_avg_sq"] for i, k in enumerate(order)}).to(dev))
            self.t = int(float(st[0]["step"]))
        else:
            self.m.zero_(); self.v.zero_(); self.t = 0
        g = sd["param_groups"][0]
        self.param_groups = [dict(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"],
                                  initial_lr=g.get("initial_lr", g["lr"]))]


class EMA:
    """Flat-buffer EMA with diffusers' warm-up decay (see module docstring: not oracle-checked)."""

    def __init__(self, model, decay=0.9999, use_ema_warmup=True, inv_gamma=1.0, power=0.75, min_decay=0.0):
        self.model, self.pstore = model, model.store
        self.decay, self.use_ema_warmup, self.inv_gamma, self.power, self.min_decay = decay, use_ema_warmup, inv_gamma, power, min_decay
        self.shadow = model.store.P.clone()
        self.optimization_step = 0
        self._backup = None

    def get_decay(self, optimization_step):
        step = max(0, optimization_step - 1)
        if step <= 0:
            return 0.0
        if self.use_ema_warmup:
            v = 1 - (1 + step / self.inv_gamma) ** -self.power
        else:
            v = (1 + step) / (10 + step)
        return max(min(v, self.decay), self.min_decay)

    def next_decay(self):
        self.optimization_step += 1
        return self.get_decay(self.optimization_step)

    def step(self, parameters=None):
        d = self.next_decay()
        self.shadow.sub_((1 - d) * (self.shadow - self.pstore.P))

    def store(self, parameters=None):           # diffusers EMAModel.store / copy_to / restore
        self._backup = self.pstore.P.clone()

    def copy_to(self, parameters=None):
        self.pstore.P.copy_(self.shadow)
        self.pstore.sync_shadow()

    def restore(self, parameters=None):
        self.pstore.P.copy_(self._backup)
        self.pstore.sync_shadow()
        self._backup = None

    def state_dict(self):
        return dict(shadow=self.shadow, optimization_step=self.optimization_step)

    def config(self):
        """The fields diffusers' EMAModel.save_pretrained adds to the model config (main_train_masked.py:119-127, 200)."""
        return dict(decay=self.decay, min_decay=self.min_decay, optimization_step=self.optimization_step, update_after_step=0,
                    use_ema_warmup=self.use_ema_warmup, inv_gamma=self.inv_gamma, power=self.power)

    def load_reference(self, sd, cfg=None):
        """Shadow parameters from a {reference key: tensor} dict (+ the counters of `config()`)."""
        self.shadow.copy_(self.pstore.flat_from_reference(sd).to(self.shadow.device))
        if cfg:
            self.optimization_step = int(cfg.get("optimization_step", self.optimization_step))
            for k in ("decay", "min_decay", "use_ema_warmup", "inv_gamma", "power"):
                if k in cfg:
                    setattr(self, k, cfg[k])


class LambdaLR:
    def __init__(self, optimizer, fn):
        self.opt, self.fn, self.k = optimizer, fn, 0
        self.base = optimizer.param_groups[0]["initial_lr"]
        optimizer.param_groups[0]["lr"] = self.base * fn(0)

    def step(self):
        self.k += 1
        self.opt.param_groups[0]["lr"] = self.base * self.fn(self.k)

    def get_last_lr(self):
        return [self.opt.param_groups[0]["lr"]]

    def state_dict(self):          # torch.optim.lr_scheduler.LambdaLR.state_dict() fields (the lambda itself is not saved)
        return dict(base_lrs=[self.base], last_epoch=self.k, _step_count=self.k + 1, _is_initial=False, _get_lr_called_within_step=False,
                    _last_lr=self.get_last_lr(), lr_lambdas=[None])

    def load_state_dict(self, sd):
        self.k = int(sd["last_epoch"])
        self.base = sd["base_lrs"][0]
        self.opt.param_groups[0]["lr"] = self.base * self.fn(self.k)


def get_lr_scheduler(name, optimizer, num_warmup_steps, num_training_steps, num_cycles=0.5):
    """'cosine' | 'hard_cosine' | 'constant' | 'linear' with warm-up (main_train_masked.py:144-165)."""
    w = max(1, num_warmup_steps)

    def warm(k):
        return k / w if k < num_warmup_steps else None
    if name == "constant":
        return LambdaLR(optimizer, lambda k: warm(k) if warm(k) is not None else 1.0)
    if name == "linear":
        return LambdaLR(optimizer, lambda k: warm(k) if warm(k) is not None else
                        max(0.0, (num_training_steps - k) / max(1, num_training_steps - num_warmup_steps)))
    if name == "cosine":
        def f(k):
            if warm(k) is not None:
                return warm(k)
            p = (k - num_warmup_steps) / max(1, num_training_steps - num_warmup_steps)
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * p)))
        return LambdaLR(optimizer, f)
    if name == "hard_cosine":
        def f(k):
            if warm(k) is not None:
                return warm(k)
            p = (k - num_warmup_steps) / max(1, num_training_steps - num_warmup_steps)
            if p >= 1.0:
                return 0.0
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((num_cycles * p) % 1.0))))
        return LambdaLR(optimizer, f)
    raise ValueError(name)


class Accelerator:
    """The subset of `accelerate.Accelerator` the trainers touch, backed by
    torch.distributed (RCCL on ROCm) when a process group is initialised."""

    def __init__(self, gradient_accumulation_steps=1, mixed_precision="bf16", device=None, split_batches=False):
        import torch.distributed as dist
        self.dist = dist if dist.is_available() and dist.is_initialized() else None
        self.num_processes = self.dist.get_world_size() if self.dist else 1
        self.process_index = self.dist.get_rank() if self.dist else 0
        self.mixed_precision = mixed_precision
        self.gradient_accumulation_steps = int(gradient_accumulation_steps)
        if self.gradient_accumulation_steps < 1:
            raise ValueError(f"gradient_accumulation_steps={gradient_accumulation_steps}")
        self.split_batches = split_batches       # accelerate default False: the LR schedule then steps num_processes times per update
        self.sync_gradients = True
        self.end_of_dataloader = False           # set by the trainer's batch loop (accelerate: by its prepared dataloader)
        self.step = 0
        self.device = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self._ckpt = {}

    @property
    def is_main_process(self):
        return self.process_index == 0

    is_local_main_process = is_main_process

    def prepare(self, *objs):
        from .unet import UNet
        for o in objs:                      # main_train_masked.py:299-307: model, optimizer, dataloader, lr_scheduler
            if isinstance(o, UNet):
                self._ckpt["model"] = o
            elif isinstance(o, AdamW):
                self._ckpt["optimizer"] = o
            elif isinstance(o, LambdaLR):
                self._ckpt["lr_scheduler"] = o
            elif isinstance(o, EMA):
                self._ckpt["ema"] = o
        return objs if len(objs) != 1 else objs[0]

    def accumulate(self, model=None):
        """`with accelerator.accumulate(model):` (ms:139) -- accelerate's `_do_sync`: the micro-step counter decides
        `sync_gradients`; the last batch of the dataloader always syncs and restarts the count.  What upstream hangs on the
        flag (scaled backward, clip, optimizer / LR step, EMA) is done by TrainStep / Trainer._step."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            if self.end_of_dataloader:
                self.step = 0
                self.sync_gradients = True
            else:
                self.step += 1
                self.sync_gradients = (self.step % self.gradient_accumulation_steps) == 0
            yield
        return ctx()

    def backward(self, loss):       # the fused train step has already produced the gradients
        return None

    def clip_grad_norm_(self, params, max_norm):
        return None

    def wait_for_everyone(self):
        if self.dist:
            self.dist.barrier()

    def print(self, *a, **k):
        if self.is_main_process:
            print(*a, **k)

    def register_for_checkpointing(self, **objs):
        """model=, optimizer=, ema=, lr_scheduler=, scheduler= : what `save_state(path)` / `load_state(path)` cover.
        (`prepare()` registers what it recognises; the trainers register the rest.)"""
        for k, v in objs.items():
            if v is not None:
                self._ckpt[k] = v

    def save_state(self, output_dir=None, model=None, optimizer=None, ema=None, **extra):
        """`accelerator.save_state(path)` (trainer_masked_mean_shift.py:267-268) in the reference's directory
        layout (mdm/checkpoint.py).  Called with the path alone, like upstream, it saves the registered objects."""
        from . import checkpoint
        c = dict(self._ckpt)
        c.update({k: v for k, v in dict(model=model, optimizer=optimizer, ema=ema).items() if v is not None})
        if c.get("model") is None:
            raise RuntimeError("Accelerator.save_state: no model registered (prepare() / register_for_checkpointing())")
        checkpoint.save_state(output_dir, c["model"], c.get("optimizer"), c.get("ema"), c.get("lr_scheduler"), c.get("scheduler"),
                              rank=self.process_index, main=self.is_main_process, extra=extra or None, step=self.step)
        self.wait_for_everyone()
        return output_dir

    def load_state(self, input_dir=None):
        """`accelerator.load_state(path)` (main_train_masked.py:268): restores every registered object in place."""
        from . import checkpoint
        c = self._ckpt
        if c.get("model") is None:
            raise RuntimeError("Accelerator.load_state: no model registered (prepare() / register_for_checkpointing())")
        extra = checkpoint.load_state(input_dir, c["model"], c.get("optimizer"), c.get("ema"), c.get("lr_scheduler"), c.get("scheduler"),
                                      rank=self.process_index)
        self.step = int(extra.get("step", self.step))          # accelerate restores its micro-step counter too
        return extra

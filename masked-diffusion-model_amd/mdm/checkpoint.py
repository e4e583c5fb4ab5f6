"""Checkpoint / resume in the reference's directory layout.

The reference saves with `accelerator.save_state(path)` (trainer_masked_mean_shift.py:267-268) through the
hooks of main_train_masked.py:195-225, and resumes with `accelerator.load_state(path)` (`resume_train`,
:250-277).  What lands in the directory is accelerate's + diffusers' format:

    <path>/unet/config.json                          model.save_pretrained (hook :203)
    <path>/unet/diffusion_pytorch_model.safetensors  state_dict in the model's own key grammar
    <path>/unet_ema/config.json                      EMAModel.save_pretrained (hook :200): the model config plus
                                                     decay, min_decay, optimization_step, update_after_step,
                                                     use_ema_warmup, inv_gamma, power
    <path>/unet_ema/diffusion_pytorch_model.safetensors   the shadow parameters under the same keys
    <path>/optimizer.bin                             torch.save(optimizer.state_dict()): torch.optim.AdamW layout,
                                                     parameters indexed in `model.parameters()` order
    <path>/scheduler.bin                             torch.save(lr_scheduler.state_dict())
    <path>/random_states_<rank>.pkl                  torch.save (despite the suffix) of {step, random_state,
                                                     numpy_random_seed, torch_manual_seed[, torch_cuda_manual_seed]}
                                                     (here also the device Philox key/offset under its own key)

optimizer.bin / scheduler.bin / random_states_<rank>.pkl are PINNED: tests/golden/train_traj.npz holds the key / shape /
dtype manifest of the directory a real `accelerate.Accelerator.save_state` wrote at the end of the reference's `train()`
run (make_golden.gen_train_traj), and tests/test_device_path_gpu.py compares what is written here against it.
`diffusers` is absent offline, so the two config.json files follow its published layout from the
call-site arguments -- parity unpinned, like the EMA decay schedule.  The tensors use the reference's 304-key
grammar and OIHW / [out,in] shapes, so `unet/` can be loaded into the reference's unet6 with
`load_state_dict(safetensors.torch.load_file(...))` and a reference checkpoint loads here.
"""
from __future__ import annotations

import json
import os
import pickle
import random

import numpy as np
import torch

WEIGHTS = "diffusion_pytorch_model.safetensors"
WEIGHTS_BIN = "diffusion_pytorch_model.bin"


def _save_tensors(sd, folder):
    from safetensors.torch import save_file
    os.makedirs(folder, exist_ok=True)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(folder, WEIGHTS))


def _load_tensors(folder):
    p = os.path.join(folder, WEIGHTS)
    if os.path.exists(p):
        from safetensors.torch import load_file
        return load_file(p)
    p = os.path.join(folder, WEIGHTS_BIN)          # older diffusers wrote a pickle
    if os.path.exists(p):
        return torch.load(p, map_location="cpu")
    raise FileNotFoundError(f"no {WEIGHTS} (or .bin) under {folder}")


def model_config(model):
    c = dict(model.cfg)
    c.update(_class_name="UNet", _library="mdm (unet6, reference code/models/unet/unet6.py)", sample_size=model.H)
    return c


def save_model(model, folder, src=None, extra_config=None):
    """`model.save_pretrained(folder)`: config.json + weights in the reference's key grammar.  `src`: another
    flat buffer with the parameter layout (the EMA shadow)."""
    sd = model.store.state_dict(order=model.reference_param_order(), src=src)
    _save_tensors(sd, folder)
    cfg = model_config(model)
    if extra_config:
        cfg.update(extra_config)
    with open(os.path.join(folder, "config.json"), "w") as fh:
        json.dump(cfg, fh, indent=2, sort_keys=True)


def load_model_tensors(folder):
    with open(os.path.join(folder, "config.json")) as fh:
        cfg = json.load(fh)
    return cfg, _load_tensors(folder)


def save_state(path, model, optimizer=None, ema=None, lr_scheduler=None, scheduler=None, rank=0, main=True, extra=None, step=0):
    os.makedirs(path, exist_ok=True)
    if main:
        save_model(model, os.path.join(path, "unet"))
        if ema is not None:
            save_model(model, os.path.join(path, "unet_ema"), src=ema.shadow, extra_config=ema.config())
        if optimizer is not None:
            torch.save(optimizer.state_dict(), os.path.join(path, "optimizer.bin"))
        if lr_scheduler is not None:
            torch.save(lr_scheduler.state_dict(), os.path.join(path, "scheduler.bin"))
    # accelerate.checkpointing.save_accelerator_state: one torch.save'd dict per rank
    states = dict(step=int(step), random_state=random.getstate(), numpy_random_seed=np.random.get_state(),
                  torch_manual_seed=torch.get_rng_state())
    if torch.cuda.is_available():
        states["torch_cuda_manual_seed"] = torch.cuda.get_rng_state_all()
    if scheduler is not None and getattr(scheduler, "dev_rng", None) is not None:
        states["mdm_philox"] = [int(v) for v in scheduler.dev_rng.dev.cpu().tolist()]
    if extra:
        states["mdm_extra"] = dict(extra)
    torch.save(states, os.path.join(path, f"random_states_{rank}.pkl"))


def load_state(path, model, optimizer=None, ema=None, lr_scheduler=None, scheduler=None, rank=0):
    """Inverse of save_state; returns the `mdm_extra` dict stored with it (or {})."""
    _cfg, sd = load_model_tensors(os.path.join(path, "unet"))
    model.load_state_dict(sd)
    if ema is not None:
        ecfg, esd = load_model_tensors(os.path.join(path, "unet_ema"))
        ema.load_reference(esd, ecfg)
    if optimizer is not None:
        optimizer.load_state_dict(torch.load(os.path.join(path, "optimizer.bin"), map_location="cpu", weights_only=False))
    if lr_scheduler is not None:
        lr_scheduler.load_state_dict(torch.load(os.path.join(path, "scheduler.bin"), map_location="cpu", weights_only=False))
    p = os.path.join(path, f"random_states_{rank}.pkl")
    extra = {}
    if os.path.exists(p):
        try:
            st = torch.load(p, map_location="cpu", weights_only=False)
        except Exception:               # round-2 checkpoints: a plain pickle
            with open(p, "rb") as fh:
                st = pickle.load(fh)
        random.setstate(st["random_state"])
        np.random.set_state(st["numpy_random_seed"])
        torch.set_rng_state(st["torch_manual_seed"])
        if "torch_cuda_manual_seed" in st and torch.cuda.is_available() and len(st["torch_cuda_manual_seed"]) == torch.cuda.device_count():
            torch.cuda.set_rng_state_all(st["torch_cuda_manual_seed"])
        extra_step = st.get("step")
        if scheduler is not None and "mdm_philox" in st:
            scheduler.dev_rng.dev.copy_(torch.tensor(st["mdm_philox"], dtype=torch.int64))
        extra = dict(st.get("mdm_extra", {}))
        if extra_step is not None:
            extra.setdefault("step", int(extra_step))
    return extra

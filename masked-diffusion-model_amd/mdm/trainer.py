"""Trainers with the reference's call surface.

`Trainer` mirrors reference trainer_masked_mean_shift.py (`Trainer.__init__` :29-66, `_run_batch`
:82-193, `_run_epoch` :196-215, `train` :218-273, `_save_ema_momentum_sample` :409-425);
`BaseTrainer` mirrors trainer_masked.py (:31-82, :95-183, :186-208, :211-272), whose constructor is
broken upstream (D2) -- the arithmetic is mean-shift with no shift, int timesteps and three return
values.  Image grids / wandb / matplotlib output of the reference are out of scope:
samples are saved as tensors.
"""
from __future__ import annotations

import os
import statistics

import torch

from .dist import GradComm
from .sampler import Sampler
from .scheduler import Scheduler
from .train_step import TrainStep


class Trainer:
    mean_shift = True

    def __init__(self, args, dataloader, dataset, dataset_hist, model, ema_model, optimizer, lr_scheduler, accelerator):
        self.args, self.dataloader, self.dataset, self.dataset_hist = args, dataloader, dataset, dataset_hist
        self.model, self.ema_model, self.optimizer, self.lr_scheduler = model, ema_model, optimizer, lr_scheduler
        self.lr_list = []
        self.accelerator = accelerator
        self.Scheduler = Scheduler(args, device=model.device)
        self.Sampler = Sampler(self.dataset, self.args, self.Scheduler, self.dataset_hist)
        self.global_step = 0
        self.timesteps_used_epoch = None
        comm = GradComm(wire=getattr(args, "grad_wire_dtype", "f32")) if getattr(accelerator, "num_processes", 1) > 1 else None
        ema = ema_model if getattr(args, "use_ema", False) else None
        self.step = TrainStep(model, self.Scheduler, args, optimizer, ema, mean_shift=self.mean_shift, comm=comm,
                              grad_accum=getattr(accelerator, "gradient_accumulation_steps", 1))
        self.loss_names = ["train_loss"]
        # what `accelerator.save_state(path)` / `load_state(path)` cover (main_train_masked.py:195-225 hooks)
        reg = getattr(accelerator, "register_for_checkpointing", None)
        if reg is not None:
            reg(model=model, optimizer=optimizer, ema=ema, lr_scheduler=lr_scheduler, scheduler=self.Scheduler)

    # ------------------------------------------------------------------------------------------
    def _batch_images(self, input):
        if "huggingface" in getattr(self.args, "dir_dataset", ""):
            return input["image"]
        return input[0]

    def _step(self, input):
        """ms:139-172.  `accelerator.accumulate` decides whether this micro-step syncs; clip + AdamW + EMA are one fused
        launch inside the step and run only then.  The LR schedule advances only on a syncing step, and -- like accelerate's
        AcceleratedScheduler, which the reference's `prepare()` at main_train_masked.py:299 wraps it in -- `num_processes`
        times per optimizer step unless `split_batches`: schedules are written in single-process steps."""
        x0 = self._batch_images(input)
        acc = self.accelerator
        with acc.accumulate(self.model):
            sync = getattr(acc, "sync_gradients", True)
            if self.Scheduler.rng_mode == "replay":
                loss = self.step.run_replay(x0, self.timesteps_used_epoch, sync=sync)
            else:
                loss = self.step.run_device(x0, self.timesteps_used_epoch, sync=sync)
            if sync:
                for _ in range(1 if getattr(acc, "split_batches", False) else max(1, getattr(acc, "num_processes", 1))):
                    self.lr_scheduler.step()
        if sync:                                  # (EMA is folded into the optimizer kernel of the step)
            self.global_step += 1
        self.learning_rate = self.lr_scheduler.get_last_lr()[0]
        self.lr_list.append(self.learning_rate)
        self.reconstruct_loss = loss
        return loss

    def _run_batch(self, batch, input, epoch, epoch_length, resume_step, dirs, visualizer):
        loss = self._step(input)
        s = self.step
        self.input, self.degraded_img, self.degrade_binary_masks = s.x0, s.x_t, s.mask
        self.shift, self.shifted_degrade_img = s.s, s.x_in
        return loss.item()                         # the reference syncs here too (ms:193)

    def _run_epoch(self, epoch, epoch_length, resume_step, dirs, visualizer):
        loss_batch = []
        self.timesteps_used_epoch = self.Scheduler.get_timesteps_epoch(epoch, epoch_length)
        for i, input in enumerate(self.dataloader, 0):
            self._mark_end_of_dataloader(i)
            loss = self._run_batch(i, input, epoch, epoch_length, resume_step, dirs, visualizer)
            if self.accelerator.is_main_process:
                loss_batch.append(loss)
        return loss_batch

    def _mark_end_of_dataloader(self, i):
        """accelerate's prepared dataloader flags its last batch, and `accumulate` syncs there whatever the micro-step
        count is (Accelerator._do_sync); ours is told by the loop."""
        n = len(self.dataloader) if hasattr(self.dataloader, "__len__") else None
        if hasattr(self.accelerator, "end_of_dataloader"):
            self.accelerator.end_of_dataloader = n is not None and i == n - 1

    def _epoch_losses(self, r):
        return r

    def train(self, epoch_start, epoch_length, resume_step, global_step, dirs, visualizer):
        a = self.args
        a.updated_ddpm_num_steps = self.Scheduler.update_ddpm_num_steps(a.ddpm_num_steps)
        self.global_step = global_step
        loss_mean_epoch = []
        self.model.train()
        for epoch in range(epoch_start, epoch_start + epoch_length):
            loss = self._epoch_losses(self._run_epoch(epoch, epoch_length, resume_step, dirs, visualizer))
            if self.accelerator.is_main_process:
                loss_mean_epoch.append(statistics.mean(loss))
            last = epoch == (epoch_start + epoch_length - 1)
            if (epoch > 0 and (epoch + 1) % a.save_images_epochs == 0) or last or \
                    (epoch + 1) % (epoch_length / a.scheduler_num_scale_timesteps) == 0:      # ms:252
                # Upstream does this block on the main process only (ms:244).  Here EVERY rank enters it: the
                # reverse sampler shards `sample_num` over the ranks and save_state ends in a barrier;
                # files are still written by the main process alone.
                if getattr(a, "use_ema", False) and getattr(a, "sampling", "momentum") == "momentum":
                    self._save_ema_momentum_sample(dirs, epoch)
                save_path = os.path.join(dirs.list_dir["checkpoint"], f"checkpoint-epoch-{epoch}")
                self.accelerator.save_state(save_path)                                           # ms:267-268
        self.loss_mean_epoch = loss_mean_epoch

    def _save_ema_momentum_sample(self, dirs, epoch):
        """ms:409-425 -- sample with the EMA weights, then put the training weights back."""
        a = self.args
        self.ema_model.store(None)
        self.ema_model.copy_to(None)
        # this rank's share of sample_num, on the sampler of record (fp32 storage, split products) unless args.sample_precision says
        # otherwise: a bf16 model's own plan is 1e-2 away from the reference sampler after 100 steps
        net = self.model.sampling_plan(self.Sampler.local_sample_num(), getattr(a, "sample_precision", "f32_split")).eval()
        sample_0, _hist = self.Sampler.sample(net, self.timesteps_used_epoch)    # gathered: [sample_num, C, H, W] on every rank
        self.ema_model.restore(None)
        self.model.train()
        self.ema_sample = sample_0
        self.ema_sample_mean = sample_0.mean()                                   # ms:421
        if self.accelerator.is_main_process:
            torch.save(sample_0.cpu(), os.path.join(dirs.list_dir["ema_sample_img"], f"ema_sample_{epoch:05d}.pt"))
        return sample_0


class BaseTrainer(Trainer):
    """trainer_masked.py surface: no dataset_hist argument, `_run_batch` -> (loss, recon_mean, degraded_mean)."""
    mean_shift = False

    def __init__(self, args, dataloader, dataset, model, ema_model, optimizer, lr_scheduler, accelerator):
        super().__init__(args, dataloader, dataset, [None] * 3, model, ema_model, optimizer, lr_scheduler, accelerator)

    def _run_batch(self, batch, input, epoch, epoch_length, resume_step, dirs, visualizer):
        loss = self._step(input)
        s = self.step
        self.input, self.degraded_img, self.degrade_binary_masks = s.x0, s.x_t, s.mask
        pred = torch.empty_like(s.x0)
        from . import ops
        m = self.model
        ops.nhwc_to_nchw(m.dt, m.y_out.data, pred, m.N, m.cout, m.H, m.W, m.cout_p)
        self.mask = pred
        self.reconstructed_img = s.x_t + pred                                               # base:126
        return loss.item(), self.reconstructed_img.mean().item(), s.x_t.mean().item()       # base:183

    def _run_epoch(self, epoch, epoch_length, resume_step, dirs, visualizer):
        out = ([], [], [])
        self.timesteps_used_epoch = self.Scheduler.get_timesteps_epoch(epoch, epoch_length)
        for i, input in enumerate(self.dataloader, 0):
            self._mark_end_of_dataloader(i)
            r = self._run_batch(i, input, epoch, epoch_length, resume_step, dirs, visualizer)
            if self.accelerator.is_main_process:
                for lst, v in zip(out, r):
                    lst.append(v)
        return out

    def _epoch_losses(self, r):
        return r[0]

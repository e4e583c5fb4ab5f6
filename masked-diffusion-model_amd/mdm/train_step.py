"""One fused masked-diffusion optimisation step on the GPU.

Arithmetic of reference trainer_masked_mean_shift.py:82-193 (mean-shift) and
trainer_masked.py:95-183 (base == mean-shift with `shift_type=non_shift`):

    t ~ U(timesteps_used_epoch) ; amount = table[t-1]                        (ms:109-112)
    x_t, m = degrade(x0, amount)                                             (scheduler.py:266-323)
    s = shift(t) ; x_in = x_t + s                                            (ms:119-120)
    pred = unet(x_in, t)                                                     (ms:140)
    loss = mean(w_n * ((x_in + pred) - s - x0)^2)                            (ms:142-159)
    backward ; clip_grad_norm_(1.0) ; AdamW ; EMA                            (ms:161-172)

Two ways to run it:
  run_replay(x0, used)  eager; all randomness drawn on the host in the reference's order
                        (parity with a reference run under the same seed);
  run_device(x0, used)  device Philox; ONE hipGraph per step on a single GPU; under data parallelism the
                        backward is cut at the gradient-bucket boundaries (front / pieces / tail graphs) and
                        each bucket's all-reduce is issued behind the piece that completes it.

Gradient accumulation (`grad_accum` > 1; `accelerator.accumulate` + `sync_gradients` upstream, ms:139-172): a
micro-step runs forward + loss (gradient scaled by 1 / grad_accum, what `accelerator.backward` does) + backward and
adds its gradient buffer to `Gacc`; the step that syncs adds `Gacc` to its own gradient, exchanges (data parallel:
one exchange per optimizer step, like DDP's no_sync), clips, updates and clears `Gacc`.
"""
from __future__ import annotations

import torch

from . import _lib, ops
from ._lib import call, ptr, stream
from .scheduler import SHIFT_KINDS, _fill_mode


class LossCell:
    """The step's loss as mdm_loss_fwd_bwd leaves it: two int64 words on the device, [0] = the mean loss in Q23.40 fixed point
    (the workgroups' partial sums meet through integer atomics, which commute: the value is bit-identical run to run), [1] =
    number of partials that were not representable.  `.item()` / `float()` synchronise and convert, like `loss.item()` upstream
    (trainer_masked_mean_shift.py:193)."""

    def __init__(self, device):
        self.raw = torch.zeros(2, device=device, dtype=torch.int64)

    def zero_(self):
        self.raw.zero_()

    def item(self):
        q, bad = self.raw.tolist()
        return float("nan") if bad else q * 2.0 ** -40

    __float__ = item


class TrainStep:
    def __init__(self, model, scheduler, args, optimizer, ema=None, mean_shift=True, comm=None, max_norm=1.0, grad_accum=1):
        self.model, self.S, self.args, self.opt, self.ema = model, scheduler, args, optimizer, ema
        self.mean_shift = mean_shift
        self.comm = comm                      # mdm.dist.GradComm or None
        if comm is not None and getattr(comm, "exchange", False) and len(getattr(model, "wgrad_groups", ())) <= 1 \
                and model.store.size * 4 > 2 * comm.bucket_bytes:
            # a model built BEFORE the process group was initialised planned ONE weight-gradient group (= one gradient bucket):
            # correct, but the whole exchange is then exposed behind the backward
            import warnings
            warnings.warn("TrainStep: data-parallel run over a model with a single weight-gradient group -- build the UNet after "
                          "mdm.dist.init_from_env() (or pass wgrad_group_bytes=32 << 20) so that gradient buckets can overlap the backward")
        self.max_norm = max_norm
        self.grad_accum = int(grad_accum)
        if self.grad_accum < 1:
            raise ValueError(f"gradient_accumulation_steps={grad_accum}")
        self.Gacc = torch.zeros_like(model.store.G) if self.grad_accum > 1 else None
        dev = model.device
        N, C, H, W = model.N, model.cin, model.H, model.W
        f = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.x0, self.x_t, self.mask, self.s, self.x_in = f(N, C, H, W), f(N, C, H, W), f(N, C, H, W), f(N, C, H, W), f(N, C, H, W)
        self.mean_pixel = f(N, C)
        self.w = f(N)
        self.loss = LossCell(dev)
        self.amount = torch.zeros(N, device=dev, dtype=torch.float64)
        self.ratio = torch.zeros(N, device=dev, dtype=torch.float64)
        self.tidx = torch.zeros(N, device=dev, dtype=torch.int32)
        self._used_key = None
        self._graphs = None
        self.use_graph = getattr(args, "use_graph", True)
        self.time_comm = False                # bench.py: record HIP events around the exposed wait for the gradient exchange
        self.comm_events = []
        self.force_cut = bool(getattr(args, "cut_step_graph", False))    # single GPU: run the data-parallel (cut) form of the step

    # ---- pieces shared by both modes -------------------------------------------------------
    def _kind(self):
        st = self.args.shift_type if self.mean_shift else "non_shift"
        if st not in SHIFT_KINDS:
            raise UnboundLocalError(f"shift_time undefined for shift_type={st!r}")
        return SHIFT_KINDS[st]

    def _emit_forward_loss(self, u, z, mask_in, Cm, weights_on):
        m, S, a = self.model, self.S, self.args
        N, C, H, W = m.N, m.cin, m.H, m.W
        fm, fc = _fill_mode(a.mean_option, a.mean_area)
        kind = self._kind()
        call("mdm_degrade", ptr(self.x0), ptr(u), ptr(mask_in), ptr(self.amount), 1, ptr(S.dev_rng.dev), 1, N, C, H * W,
             C if mask_in is not None else Cm, fm, fc, ptr(self.x_t), ptr(self.mask), ptr(self.mean_pixel), stream())
        per_col = int(S.reference_quirks and kind in (3, 4) and N == W and N > 1)
        call("mdm_shift", ptr(self.x_t), ptr(z), ptr(self.ratio), ptr(S.dev_rng.dev), 2, kind,
             float(getattr(a, "noise_mean", 0.0)), per_col, N, C, H, W, ptr(self.s), ptr(self.x_in), m.dt,
             ptr(m.x_in.data), m.cin_p, stream())
        m.forward_plan.run() if _lib._recording is None else _lib._recording.extend(m.forward_plan)
        call("mdm_loss_fwd_bwd", m.dt, ptr(m.y_out.data), ptr(self.x_in), ptr(self.s) if kind != 0 else None, ptr(self.x0),
             ptr(self.w) if weights_on else None, N, C, H, W, m.cout_p, 1.0 / self.grad_accum, ptr(m.y_out.grad), ptr(self.loss.raw), stream())

    def _hyper(self):
        d = self.ema.next_decay() if self.ema is not None else 0.0
        self.opt.hyper(ema_decay=d)

    # ---- replay mode (parity) --------------------------------------------------------------
    def run_replay(self, x0, used, sync=True):
        """Eager step with the reference's host RNG order. Returns the loss cell.  `sync`: this micro-step
        ends in the optimizer update (always, unless grad_accum > 1)."""
        m, S, a = self.model, self.S, self.args
        N, C, H, W = m.N, m.cin, m.H, m.W
        dev = m.device
        self.x0.copy_(x0.to(torch.float32))
        # Data-parallel replay (`Scheduler.replay_rows = (lo, hi, n)`): the reference seeds every rank alike, so each rank draws the
        # host numbers of the WHOLE n-sample batch and keeps its rows -- the ranks together see exactly what one process with n
        # samples draws (tests/test_ema_dp_gpu.py compares the two).
        nh, rows = S._host_rows(N)
        timeindex = torch.randint(low=0, high=len(used), size=(nh,))                      # ms:109
        t_all = torch.index_select(torch.tensor(used), 0, timeindex)
        t_all = t_all.to(torch.float32) if self.mean_shift else t_all                     # ms:110 / base:115
        timeindex, t = timeindex[rows], t_all[rows]
        amount = S.get_black_area_num_pixels_time(t.to(dev))                              # ms:112
        weights_on = bool(getattr(a, "loss_weight_use", False))
        if weights_on:
            self.w.copy_(S.get_weight_timesteps(timeindex, a.loss_weight_power_base))
        u = mask_in = None
        Cm = 1
        if a.select_degrade_pixel == "indexing":
            if amount.dtype.is_floating_point:
                raise TypeError("indexing needs integer pixel counts (D7)")
            amount_all = S.get_black_area_num_pixels_time(t_all.to(dev)) if nh != N else amount
            mk = torch.ones(nh, H * W)
            for i, num in enumerate(amount_all.cpu()):
                mk[i, torch.randperm(H * W)[:num]] = 0.0                                  # scheduler.py:281-282
            mask_in = mk[rows].reshape(N, 1, H, W).expand(N, C, H, W).contiguous().to(dev)
        else:
            Cm = S._check_degrade_args(self.x0)
            self.amount.copy_(amount)
            u = torch.empty(nh, Cm * H * W).uniform_(0.0, 1.0)[rows].contiguous().to(dev)  # scheduler.py:288/294
        kind = self._kind()
        z = None
        if kind != 0:
            self.ratio.copy_(torch.index_select(S.ratio_dev, 0, (t.int() - 1).to(dev)))
            ratio_all = torch.index_select(S.ratio_list, 0, (t_all.int() - 1).long()) if nh != N else self.ratio.cpu()
            z = S._shift_draws(nh, C, H, W, ratio_all)[rows].to(dev).contiguous()
        m.t_in.copy_(t.to(torch.float32))
        self.loss.zero_()                   # (the device path clears it in its first launch, mdm_draw_timesteps)
        self._emit_forward_loss(u, z, mask_in, Cm, weights_on)
        m.store.G.zero_()
        m.run_backward()
        self.last = dict(timeindex=timeindex, t=t)
        if self.grad_accum > 1 and not sync:
            ops.add_(_lib.F32, self.Gacc, m.store.G)
            return self.loss
        if self.grad_accum > 1:
            ops.add_(_lib.F32, m.store.G, self.Gacc)
        self._finish_update()
        if self.grad_accum > 1:
            ops.fill(self.Gacc, 0.0)
        return self.loss

    def _finish_update(self):
        gmul = 1.0
        if self.comm is not None:
            self.comm.allreduce_all(self.model.store.G)
            gmul = 1.0 / self.comm.world
        self._hyper()
        self.opt.emit_update(self.ema.shadow if self.ema is not None else None, self.max_norm, gmul)

    # ---- device mode (fast path) -----------------------------------------------------------
    def _upload_used(self, used):
        key = (len(used), used[0], used[-1], used[len(used) // 2])
        if key == self._used_key:
            return
        S, a = self.S, self.args
        dev = self.model.device
        self.used_dev = torch.tensor(used, dtype=torch.int32, device=dev)
        sel = a.select_degrade_pixel
        if sel == "indexing":
            if not isinstance(S.black_area_pixels, torch.Tensor):
                self.table_dev = S.pixels_dev.to(torch.float64)
            else:
                raise TypeError("indexing needs integer pixel counts (D7)")
        else:
            self.table_dev = S.ratio_dev
        self.wtab_dev = None
        if getattr(a, "loss_weight_use", False):
            alpha = torch.linspace(start=1, end=0, steps=S.updated_ddpm_num_steps)
            self.wtab_dev = torch.pow(a.loss_weight_power_base, alpha)[:len(used)].to(dev)
        if self._used_key is not None:
            self._graphs = None          # tables moved: re-capture
        self._used_key = key

    def _emit_device_front(self):
        m, S, a = self.model, self.S, self.args
        N, C, H, W = m.N, m.cin, m.H, m.W
        rng = ptr(S.dev_rng.dev)
        n_used = self.used_dev.numel()
        S.dev_rng.advance()                 # first launch of the step (part of the captured graph): a fresh Philox offset
        call("mdm_draw_timesteps", rng, ptr(self.used_dev), n_used, ptr(self.table_dev), ptr(self.wtab_dev), N,
             ptr(m.t_in), ptr(self.amount), ptr(self.w), ptr(self.tidx), ptr(S.ratio_dev), ptr(self.ratio), ptr(self.loss.raw), stream())
        mask_in, Cm = None, 1
        if a.select_degrade_pixel == "indexing":
            call("mdm_index_mask", ptr(self.amount), 1, rng, 1, N, C, H * W, ptr(self.mask), stream())
            mask_in = self.mask
        else:
            Cm = S._check_degrade_args(self.x0)
        self._emit_forward_loss(None, None, mask_in, Cm, self.wtab_dev is not None)
        m.emit_zero_grad()

    def _build_graphs(self):
        """front = draws + degrade + shift + forward + loss + gradient zeroing; backward; tail = optimizer.
        Single GPU, no accumulation: the whole step is ONE graph.  Data parallel: the backward launch list is cut behind
        every grouped weight-gradient launch that completes a gradient bucket (`GradComm.plan_chunks`), one graph per piece,
        and the bucket's all-reduce is issued behind its piece -- RCCL then runs on its own stream under the next piece.
        (Running the grouped weight gradients themselves on a second stream was measured slower on this stack -- DESIGN
        finding 16 -- and is gone.)  Accumulation: front / whole backward / tail graphs with the adds in between."""
        m = self.model
        with _lib.Recording() as front:
            self._emit_device_front()
        with _lib.Recording() as tail:
            gmul = 1.0 / self.comm.world if self.comm is not None else 1.0
            self.opt.emit_update(self.ema.shadow if self.ema is not None else None, self.max_norm, gmul)
        mk = (lambda r: _lib.GraphExec(r)) if self.use_graph else (lambda r: r)

        def rec(cs):
            r = _lib.Recording()
            r.calls, r.keep = list(cs), m.backward_plan.keep
            return r
        calls = m.backward_plan.calls
        if self.grad_accum > 1:
            self._graphs = ("accum", mk(front), mk(rec(calls)), mk(tail))
            return
        if self.comm is None and not self.force_cut:
            whole = _lib.Recording()
            whole.extend(front); whole.extend(m.backward_plan); whole.extend(tail)
            self._graphs = ("whole", mk(whole))
            return
        # cut form: pieces end where a gradient bucket becomes complete (a cut index counts launches of the backward plan).
        # `cut_step_graph` on a single GPU plans the same buckets a data-parallel run would (world = 1: no exchange), so
        # that the price of cutting the step graph can be measured without a second GPU (bench.py --cut-graph)
        from .dist import GradComm
        planner = self.comm if self.comm is not None else GradComm()
        cuts = sorted(set(min(c, len(calls)) for c in planner.plan_chunks(m)))
        pieces, lo = [], 0
        for b, c in enumerate(cuts):
            pieces.append((mk(rec(calls[lo:c])) if c > lo else None, b))
            lo = c
        rest = mk(rec(calls[lo:])) if lo < len(calls) else None
        self._graphs = ("cut", mk(front), pieces, rest, mk(tail))

    def run_device(self, x0, used, sync=True):
        """Device-RNG step as hipGraph replays.  `x0` None = reuse the batch already in `self.x0`.  `sync`: see run_replay."""
        if x0 is not None:
            self.x0.copy_(x0.to(torch.float32), non_blocking=True)
        self._upload_used(used)
        if self._graphs is None:
            self._build_graphs()
        go = (lambda g: g.launch()) if self.use_graph else (lambda g: g.run())
        kind = self._graphs[0]
        G = self.model.store.G
        if kind == "accum":
            _, front, bwd, tail = self._graphs
            go(front); go(bwd)
            if not sync:
                ops.add_(_lib.F32, self.Gacc, G)
                return self.loss
            ops.add_(_lib.F32, G, self.Gacc)
            if self.comm is not None:
                self.comm.allreduce_all(G)
            self._hyper()
            go(tail)
            ops.fill(self.Gacc, 0.0)
            return self.loss
        self._hyper()
        if kind == "whole":
            go(self._graphs[1])
            return self.loss
        _, front, pieces, rest, tail = self._graphs
        go(front)
        for g, b in pieces:
            if g is not None:
                go(g)
            if self.comm is not None:
                self.comm.reduce_bucket(b, G)
        if rest is not None:
            go(rest)
        if self.comm is not None:
            if self.time_comm:
                # what the compute stream WAITS for the exchange behind the last backward piece (the part of the all-reduce that
                # the backward did not hide): HIP events on the compute stream around the stream-level waits
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.comm.wait_all(G)
                e1.record()
                self.comm_events.append((e0, e1))
            else:
                self.comm.wait_all(G)
        go(tail)
        return self.loss

    def comm_report(self):
        """dict for bench.py's `comm` object: bytes on the wire per step, bucket plan, and -- when `time_comm` was set -- the
        exposed wait per step (median / max over the recorded steps; synchronises)."""
        c, m = self.comm, self.model
        if c is None:
            return None
        out = {"world": c.world, "wire": c.wire, "bytes_per_step": int(m.store.size * (2 if c.wire == "bf16" else 4)),
               "buckets": len(c.buckets), "bucket_bytes": int(c.bucket_bytes), "tail_bytes": int(c.tail_bytes),
               "bucket_sizes_mb": [round((hi - lo) * 4 / 2 ** 20, 1) for lo, hi in c.buckets],
               "wgrad_groups": len(m.wgrad_groups), "wgrad_group_bytes": int(min(m.wgrad_group_bytes, m.store.size * 4))}
        if self.comm_events:
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in self.comm_events)
            out.update(exposed_ms=round(ms[len(ms) // 2], 4), exposed_ms_max=round(ms[-1], 4), exposed_samples=len(ms))
            self.comm_events = []
        return out

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
  run_device(x0, used)  device Philox; the step is three hipGraphs (forward+loss, backward chunks,
                        optimizer) so that data-parallel ranks can all-reduce gradient buckets
                        between backward chunks while the next chunk runs.
"""
from __future__ import annotations

import torch

from . import _lib, ops
from ._lib import call, ptr, stream
from .scheduler import SHIFT_KINDS, _fill_mode


class TrainStep:
    def __init__(self, model, scheduler, args, optimizer, ema=None, mean_shift=True, comm=None, max_norm=1.0):
        self.model, self.S, self.args, self.opt, self.ema = model, scheduler, args, optimizer, ema
        self.mean_shift = mean_shift
        self.comm = comm                      # mdm.dist.GradComm or None
        self.max_norm = max_norm
        dev = model.device
        N, C, H, W = model.N, model.cin, model.H, model.W
        f = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.x0, self.x_t, self.mask, self.s, self.x_in = f(N, C, H, W), f(N, C, H, W), f(N, C, H, W), f(N, C, H, W), f(N, C, H, W)
        self.mean_pixel = f(N, C)
        self.w = f(N)
        self.loss = f(1)
        self.amount = torch.zeros(N, device=dev, dtype=torch.float64)
        self.ratio = torch.zeros(N, device=dev, dtype=torch.float64)
        self.tidx = torch.zeros(N, device=dev, dtype=torch.int32)
        self._used_key = None
        self._graphs = None
        self.use_graph = getattr(args, "use_graph", True)
        self.overlap = bool(getattr(args, "overlap_wgrads", False))      # grouped weight gradients on a second stream
        self.side = None

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
             ptr(self.w) if weights_on else None, N, C, H, W, m.cout_p, 1.0, ptr(m.y_out.grad), ptr(self.loss), stream())

    def _hyper(self):
        d = self.ema.next_decay() if self.ema is not None else 0.0
        self.opt.hyper(ema_decay=d)

    # ---- replay mode (parity) --------------------------------------------------------------
    def run_replay(self, x0, used):
        """Eager step with the reference's host RNG order. Returns the loss tensor."""
        m, S, a = self.model, self.S, self.args
        N, C, H, W = m.N, m.cin, m.H, m.W
        dev = m.device
        self.x0.copy_(x0.to(torch.float32))
        timeindex = torch.randint(low=0, high=len(used), size=(N,))                       # ms:109
        t = torch.index_select(torch.tensor(used), 0, timeindex)
        t = t.to(torch.float32) if self.mean_shift else t                                 # ms:110 / base:115
        amount = S.get_black_area_num_pixels_time(t.to(dev))                              # ms:112
        weights_on = bool(getattr(a, "loss_weight_use", False))
        if weights_on:
            self.w.copy_(S.get_weight_timesteps(timeindex, a.loss_weight_power_base))
        u = mask_in = None
        Cm = 1
        if a.select_degrade_pixel == "indexing":
            if amount.dtype.is_floating_point:
                raise TypeError("indexing needs integer pixel counts (D7)")
            mk = torch.ones(N, H * W)
            for i, num in enumerate(amount.cpu()):
                mk[i, torch.randperm(H * W)[:num]] = 0.0                                  # scheduler.py:281-282
            mask_in = mk.reshape(N, 1, H, W).expand(N, C, H, W).contiguous().to(dev)
        else:
            Cm = S._check_degrade_args(self.x0)
            self.amount.copy_(amount)
            u = torch.empty(N, Cm * H * W).uniform_(0.0, 1.0).to(dev)                     # scheduler.py:288/294
        kind = self._kind()
        z = None
        if kind != 0:
            self.ratio.copy_(torch.index_select(S.ratio_dev, 0, (t.int() - 1).to(dev)))
            z = S._shift_draws(N, C, H, W, self.ratio.cpu()).to(dev).contiguous()
        m.t_in.copy_(t.to(torch.float32))
        self.loss.zero_()                   # (the device path clears it in its first launch, mdm_draw_timesteps)
        self._emit_forward_loss(u, z, mask_in, Cm, weights_on)
        m.store.G.zero_()
        m.run_backward()
        self._finish_update()
        self.last = dict(timeindex=timeindex, t=t)
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
             ptr(m.t_in), ptr(self.amount), ptr(self.w), ptr(self.tidx), ptr(S.ratio_dev), ptr(self.ratio), ptr(self.loss), stream())
        mask_in, Cm = None, 1
        if a.select_degrade_pixel == "indexing":
            call("mdm_index_mask", ptr(self.amount), 1, rng, 1, N, C, H * W, ptr(self.mask), stream())
            mask_in = self.mask
        else:
            Cm = S._check_degrade_args(self.x0)
        self._emit_forward_loss(None, None, mask_in, Cm, self.wtab_dev is not None)
        m.emit_zero_grad()

    def _build_graphs(self):
        """front = draws + degrade + shift + forward + loss; the backward is cut at every grouped weight-gradient launch
        into [chain graph, group] pieces; tail = optimizer.  With `overlap` the groups are not part of any graph: each is
        issued on a second stream as soon as the chain piece that produced its operands has been enqueued, so it runs
        NEXT TO the following chain piece (hipGraph branches were measured NOT to run concurrently on this stack; two
        streams do).  Data parallel: the bucket of a piece is all-reduced once its group has finished."""
        m = self.model
        with _lib.Recording() as front:
            self._emit_device_front()
        calls = m.backward_plan.calls
        is_group = lambda c: c[0] == "mdm_wgrad_group_launch"
        # pieces: (chain calls, group call or None); bucket_after[j] = index of the bucket that is complete after piece j
        pieces, lo = [], 0
        for i, c in enumerate(calls):
            if is_group(c):
                pieces.append((calls[lo:i], c))
                lo = i + 1
        pieces.append((calls[lo:], None))
        ends, pos = [], 0                       # number of calls consumed after each piece (group call included)
        for chain, grp in pieces:
            pos += len(chain) + (1 if grp is not None else 0)
            ends.append(pos)
        cuts = self.comm.plan_chunks(m) if self.comm is not None else []
        self.bucket_after = {}
        for b, c in enumerate(cuts):            # a bucket is complete after the first piece that ends at or behind its cut
            j = next(k for k, e in enumerate(ends) if e >= c)
            self.bucket_after.setdefault(j, []).append(b)
        with _lib.Recording() as tail:
            gmul = 1.0 / self.comm.world if self.comm is not None else 1.0
            self.opt.emit_update(self.ema.shadow if self.ema is not None else None, self.max_norm, gmul)
        mk = (lambda r: _lib.GraphExec(r)) if self.use_graph else (lambda r: r)

        def rec(cs):
            r = _lib.Recording()
            r.calls, r.keep = list(cs), m.backward_plan.keep
            return r
        if self.comm is None and not self.overlap:
            # single GPU, serial: nothing happens between the pieces, so the whole step is ONE graph
            whole = _lib.Recording()
            whole.extend(front); whole.extend(m.backward_plan); whole.extend(tail)
            self._graphs = (mk(whole), [], None)
            return
        if not self.overlap:                    # groups stay inside the chain graphs
            built = [(mk(rec(list(chain) + ([grp] if grp is not None else []))), None) for chain, grp in pieces]
        else:
            built = [(mk(rec(chain)) if chain else None, grp) for chain, grp in pieces]
            if self.side is None:
                self.side = torch.cuda.Stream()
                self.ev_main, self.ev_side = torch.cuda.Event(), torch.cuda.Event()
        self._graphs = (mk(front), built, mk(tail))

    def run_device(self, x0, used):
        """Device-RNG step as hipGraph replays.  `x0` None = reuse the batch already in `self.x0`."""
        if x0 is not None:
            self.x0.copy_(x0.to(torch.float32), non_blocking=True)
        self._upload_used(used)
        if self._graphs is None:
            self._build_graphs()
        front, pieces, tail = self._graphs
        self._hyper()
        go = (lambda g: g.launch()) if self.use_graph else (lambda g: g.run())
        go(front)
        if tail is None:
            return self.loss
        main = torch.cuda.current_stream()
        forked = False
        for j, (chain, grp) in enumerate(pieces):
            if chain is not None:
                go(chain)
            if grp is not None:                 # overlap: the group goes to the second stream, behind this chain piece
                self.ev_main.record(main)
                self.side.wait_event(self.ev_main)
                _lib.check(grp[1](*grp[2], self.side.cuda_stream), grp[0])
                forked = True
            for b in (self.bucket_after.get(j, ()) if self.comm is not None else ()):
                if grp is not None:
                    with torch.cuda.stream(self.side):      # the exchange waits for the group, not for the chain
                        self.comm.reduce_bucket(b, self.model.store.G)
                else:
                    self.comm.reduce_bucket(b, self.model.store.G)
        if forked:
            self.ev_side.record(self.side)
            main.wait_event(self.ev_side)
        if self.comm is not None:
            self.comm.wait_all(self.model.store.G)
        go(tail)
        return self.loss

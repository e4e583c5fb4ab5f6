"""Data-parallel gradient exchange: one process per GPU, torch.distributed ("nccl" = RCCL over
xGMI on ROCm; "gloo" for the CPU tests), bucketed all-reduce of the flat gradient buffer
overlapped with the remaining backward chunks.

Replaces what accelerate/DDP insert around `accelerator.backward(loss)` (reference
trainer_masked_mean_shift.py:161, main_train_masked.py:299).  Differences by design:
no per-step barrier (the reference's `wait_for_everyone()` at :183 serialises steps), gradients
are SUMMED on the wire and divided by `world` inside the optimizer kernel, and because the
gradients already live in one flat buffer a bucket is just a contiguous slice: no copies.

Bucket sizing for MI355X: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so RCCL's ring
is per-link bound; 143 MB of fp32 gradients in ~32 MB buckets keeps 4-5 collectives in flight
behind a ~1 ms backward while staying far above the latency-bound regime (<1 MB).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def init_from_env(backend=None, single=False):
    """Initialise the default process group from torchrun's env (RANK/WORLD_SIZE/MASTER_*).  `single`: also for a world of
    one (a rehearsal of the RCCL path on a one-GPU box; a one-rank job needs no group otherwise)."""
    import os
    if dist.is_initialized():
        return
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and not single:
        return
    if single:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(k, v)
        if "MASTER_PORT" not in os.environ:        # a free port: two rehearsals on one host must not meet on a fixed one
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl" and os.environ.get("MDM_FORCE_DEVICE") is None:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend)


class GradComm:
    """wire="f32": all-reduce the fp32 gradient slices in place (what accelerate/DDP do for the reference: fp32 master
    gradients, bit-identical replicas).  wire="bf16": each bucket is rounded to bf16 into a staging buffer, all-reduced
    there and widened back -- half the bytes on xGMI (71.5 MB instead of 143 MB per step at cfg2), at the
    price of one rounding of every gradient element before the sum (replicas stay bit-identical: every rank widens the
    same reduced values).  Default fp32: at cfg2 the exchange hides behind the backward except for the last bucket."""

    def __init__(self, group=None, bucket_bytes=32 << 20, tail_bytes=4 << 20, wire="f32", always_exchange=False):
        assert wire in ("f32", "bf16"), wire
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a world of one has nothing to exchange; `always_exchange` issues the collectives anyway (the sum over one rank is
        # the identity) so that a one-GPU box can run the RCCL code path: tests/test_dp_gpu.py, bench.py MDM_REHEARSE_COMM=1
        self.exchange = self.world > 1 or (always_exchange and dist.is_initialized())
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bucket_bytes = bucket_bytes
        self.tail_bytes = min(tail_bytes, bucket_bytes)
        self.wire = wire
        self.buckets = []        # (lo, hi) element ranges of the flat gradient buffer, in completion order
        self._handles = []
        self._stage = None       # bf16 staging copy of the flat gradient buffer (wire="bf16")

    # ---- planning --------------------------------------------------------------------------
    @staticmethod
    def plan_buckets(marks, total, bucket_elems, tail_elems=None):
        """marks: [(n_calls_after_piece, lowest_param_offset_complete)] in backward order (offsets
        non-increasing, the last one 0).  -> (cuts, buckets): after call index cuts[i], the gradient
        slice buckets[i] = (lo, hi) is final.  Buckets tile [0, total) from the top down.
        `tail_elems`: once less than two buckets are left, a bucket closes at half of what is left (never
        below tail_elems): only the LAST exchange is exposed behind the backward, so it should be small."""
        cuts, buckets = [], []
        hi = total
        for n_calls, lo in marks:
            last = lo == 0
            want = bucket_elems
            if tail_elems is not None and hi < 2 * bucket_elems:
                want = max(tail_elems, hi // 2)
            if (hi - lo >= want) or (last and hi > lo):
                if cuts and cuts[-1] == n_calls:       # no new calls since the previous cut: widen it
                    buckets[-1] = (lo, buckets[-1][1])
                else:
                    cuts.append(n_calls)
                    buckets.append((lo, hi))
                hi = lo
        assert hi == 0 and buckets, "bucket plan must end at offset 0"
        return cuts, buckets

    def plan_chunks(self, model):
        marks = list(model.bwd_marks)
        n_total = len(model.backward_plan.calls)
        marks[-1] = (n_total, 0)
        cuts, self.buckets = self.plan_buckets(marks, model.store.size, self.bucket_bytes // 4, tail_elems=self.tail_bytes // 4)
        cuts[-1] = n_total
        return cuts

    # ---- exchange ---------------------------------------------------------------------------
    def reduce_bucket(self, i, G):
        if not self.exchange:
            return
        lo, hi = self.buckets[i]
        if self.wire == "bf16":
            if self._stage is None or self._stage.numel() != G.numel():
                self._stage = torch.empty(G.numel(), dtype=torch.bfloat16, device=G.device)
            self._Gref = G
            st = self._stage[lo:hi]
            st.copy_(G[lo:hi])                                   # round once, on the compute stream, behind the bucket's kernels
            h = dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._handles.append((h, lo, hi))
            return
        self._handles.append((dist.all_reduce(G[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None))

    def wait_all(self, G=None):
        for h, lo, hi in self._handles:
            h.wait()             # on NCCL/RCCL: stream-level wait, the host does not block
            if lo is not None:
                (G if G is not None else self._Gref)[lo:hi].copy_(self._stage[lo:hi])       # widen the reduced bucket back
        self._handles = []

    def allreduce_all(self, G):
        if self.exchange:
            dist.all_reduce(G, op=dist.ReduceOp.SUM, group=self.group)

"""ctypes binding of libmdm_hip.so (include/mdm_hip.h).

The product path has NO fallback: if the shared library is missing, or a call
returns non-zero, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

F32, BF16 = 0, 1
_HERE = os.path.dirname(os.path.abspath(__file__))
# (MDM_LIB_PATH: load another BUILD of the same library -- A/B timing of two builds on one box; it selects a file, not a code path)
LIB_PATH = os.environ.get("MDM_LIB_PATH") or os.path.join(_HERE, "libmdm_hip.so")

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class GemmDesc(C.Structure):
    """Mirror of `mdm_gemm_desc` (include/mdm_hip.h)."""
    _fields_ = [
        ("dtype", i32), ("layout", i32), ("M", i32), ("N", i32), ("K", i32), ("batch", i32),
        ("sA", i64), ("sB", i64), ("sD", i64), ("sR", i64),
        ("A", vp), ("lda", i32), ("f32_split", i32),
        ("B", vp), ("ldb", i32), ("_p1", i32),
        ("conv", i32), ("OH", i32), ("OW", i32), ("IH", i32), ("IW", i32),
        ("KH", i32), ("KW", i32), ("stride", i32), ("pad_t", i32), ("pad_l", i32), ("transposed", i32), ("ups", i32),
        ("C0", i32), ("C1", i32), ("Ck", i32),
        ("src0", vp), ("src1", vp), ("ld0", i32), ("ld1", i32), ("wtap", i64),
        ("D0", vp), ("D1", vp), ("ldd0", i32), ("ldd1", i32), ("N0", i32), ("out_f32", i32), ("alpha", f32),
        ("acc0", i32), ("acc1", i32), ("bias", vp), ("rowvec", vp), ("rv_ld", i32), ("rows_per_img", i32),
        ("resid", vp), ("ldr", i32), ("splitk", i32), ("dtap", i64), ("ws", vp), ("ws_bytes", i64), ("dbias", vp),
        ("gnb_x", vp), ("gnb_stats", vp), ("gnb_gamma", vp), ("gnb_beta", vp), ("gnb_dgamma", vp), ("gnb_dbeta", vp),
        ("gnb_sum_img", vp), ("gnb_sum_all", vp), ("gnb_G", i32), ("gnb_silu", i32), ("gnb_sum_ld", i32), ("_p3", i32), ("gnb_add", vp),
        ("gnf_out", vp), ("gnf_gamma", vp), ("gnf_beta", vp), ("gnf_stats", vp), ("gnf_G", i32), ("gnf_silu", i32), ("gnf_eps", f32), ("_p4", i32),
        ("B_split", vp),
    ]


_PROTOS = {
    "mdm_version": ([], i32),
    "mdm_device_count": ([], i32),
    "mdm_gemm": ([C.POINTER(GemmDesc), vp], i32),
    "mdm_gemm_pair": ([C.POINTER(GemmDesc), C.POINTER(GemmDesc), vp], i32),
    "mdm_wgrad_group_accepts": ([C.POINTER(GemmDesc)], i32),
    "mdm_wgrad_group_create": ([C.POINTER(GemmDesc), i32, vp, i64, C.POINTER(i64), C.POINTER(vp)], i32),
    "mdm_wgrad_group_launch": ([vp, vp], i32),
    "mdm_wgrad_group_destroy": ([vp], i32),
    "mdm_chain_accepts": ([C.POINTER(GemmDesc), C.POINTER(GemmDesc)], i32),
    "mdm_chain_create": ([C.POINTER(GemmDesc), C.POINTER(i32), i32, vp, i64, C.POINTER(i64), C.POINTER(vp)], i32),
    "mdm_chain_launch": ([vp, vp], i32),
    "mdm_chain_status": ([vp, C.POINTER(C.c_uint32)], i32),
    "mdm_chain_destroy": ([vp], i32),
    "mdm_gemm_plan": ([C.POINTER(GemmDesc), C.POINTER(i32), C.POINTER(i64)], i32),
    "mdm_gemm_can_fuse_gn_bwd": ([C.POINTER(GemmDesc), i32], i32),
    "mdm_gemm_can_fuse_gn_fwd": ([C.POINTER(GemmDesc), i32], i32),
    "mdm_groupnorm_fwd": ([i32, vp, i32, vp, i32, i32, i32, i32, f32, vp, vp, i32, vp, vp, vp, vp], i32),
    "mdm_groupnorm_bwd": ([i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp], i32),
    "mdm_groupnorm_bwd_sums": ([i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, vp, vp, vp], i32),
    "mdm_groupnorm_bwd_ws_floats": ([i32, i32, i32], i64),
    "mdm_groupnorm_bwd_add": ([i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp], i32),
    "mdm_attn_supported": ([i32, i32, i32], i32),
    "mdm_attn_fwd": ([i32, vp, vp, vp, i32, i32, i32, f32, vp], i32),
    "mdm_attn_f32_small_supported": ([i32, i32], i32),
    "mdm_attn_f32_small_fwd": ([vp, vp, vp, i32, i32, i32, f32, vp], i32),
    "mdm_attn_bwd": ([i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp], i32),
    "mdm_attn_mh_fwd": ([i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp], i32),
    "mdm_attn_mh_bwd": ([i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp], i32),
    "mdm_softmax_fwd": ([i32, vp, i32, i32, vp], i32),
    "mdm_softmax_bwd": ([i32, vp, vp, i32, i32, vp], i32),
    "mdm_timestep_embedding": ([vp, i32, i32, vp, vp], i32),
    "mdm_timestep_embedding2": ([vp, i32, i32, i32, f32, vp, vp], i32),
    "mdm_silu_fwd": ([vp, vp, i64, vp], i32),
    "mdm_silu_bwd": ([vp, vp, vp, i32, i64, vp], i32),
    "mdm_skinny_supported": ([i32, i32, i32, i32], i32),
    "mdm_skinny_linear_fwd": ([vp, i32, vp, i32, f32, vp, vp, i32, vp, i32, i32, i32, vp, i32, vp, vp], i32),
    "mdm_skinny_linear_bwd": ([vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp], i32),
    "mdm_silu_bwd_sum": ([vp, vp, i32, i64, vp, vp], i32),
    "mdm_colsum": ([i32, vp, i32, i32, i32, vp, i32, i32, vp, vp], i32),
    "mdm_sumpool2": ([i32, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "mdm_add": ([i32, vp, vp, i64, vp], i32),
    "mdm_add3": ([i32, vp, vp, vp, i64, vp], i32),
    "mdm_nchw_to_nhwc": ([i32, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "mdm_nhwc_to_nchw": ([i32, vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "mdm_draw_timesteps": ([vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp], i32),
    "mdm_degrade": ([vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp], i32),
    "mdm_index_mask": ([vp, i32, vp, i32, i32, i32, i32, vp, vp], i32),
    "mdm_shift": ([vp, vp, vp, vp, i32, i32, f32, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp], i32),
    "mdm_zero_pad_channels": ([i32, vp, i64, i32, i32, vp], i32),
    "mdm_loss_fwd_bwd": ([i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp], i32),
    "mdm_sampler_x0": ([i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp], i32),
    "mdm_sampler_update": ([vp, vp, vp, vp, i32, i64, vp], i32),
    "mdm_normalize01": ([vp, vp, i32, i32, vp], i32),
    "mdm_unit_rows": ([vp, vp, i32, i32, f32, vp], i32),
    "mdm_col_argmax": ([vp, i32, i32, vp, vp, vp], i32),
    "mdm_rng_advance": ([vp, vp], i32),
    "mdm_sampler_step_params": ([vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp], i32),
    "mdm_sqnorm": ([vp, i64, vp, vp], i32),
    "mdm_adamw_ema": ([vp, vp, vp, vp, vp, vp, i64, vp, vp, f32, f32, vp], i32),
    "mdm_cast_bf16": ([vp, vp, i64, vp], i32),
    "mdm_transpose_shadow_bf16": ([vp, vp, vp, i32, vp], i32),
    "mdm_split_shadow": ([vp, vp, vp, i32, vp], i32),
    "mdm_fill_f32": ([vp, f32, i64, vp], i32),
    "mdm_fill_segments_f32": ([vp, vp, i32, f32, vp], i32),
    "mdm_graph_begin": ([vp], i32),
    "mdm_graph_end": ([vp, C.POINTER(vp)], i32),
    "mdm_graph_launch": ([vp, vp], i32),
    "mdm_graph_destroy": ([vp], i32),
    "mdm_event_create": ([C.POINTER(vp)], i32),
    "mdm_event_record": ([vp, vp], i32),
    "mdm_event_elapsed_ms": ([vp, vp, C.POINTER(f32)], i32),
    "mdm_event_destroy": ([vp], i32),
    "mdm_stream_sync": ([vp], i32),
}

EXPORTS = ["mdm_last_error"] + list(_PROTOS)

_lib = None


def load():
    """Load the library once; raise loudly if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C masked-diffusion-model_amd/csrc`).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.mdm_last_error.restype = C.c_char_p
    lib.mdm_last_error.argtypes = []
    for name, (args, res) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mdm_last_error().decode()
        raise RuntimeError(f"libmdm_hip {what} failed ({rc}): {msg}")


class Recording:
    """A recorded launch sequence: every C-ABI call made while it is active is appended
    (function, args-without-stream) instead of being executed; `run()` replays the list on
    the current stream.  All kernel entry points take the stream as their LAST argument."""

    def __init__(self):
        self.calls = []
        self.keep = []      # objects (descriptors, tensors) that must outlive the list
        self.flops = {}     # call index -> (algorithmic FLOPs, dtype) for contraction launches

    def __enter__(self):
        global _recording
        self._prev = _recording
        _recording = self
        return self

    def __exit__(self, *exc):
        global _recording
        _recording = self._prev
        return False

    def run(self, st=None):
        st = stream() if st is None else st
        for name, fn, args in self.calls:
            rc = fn(*args, st)
            if rc != 0:
                check(rc, name)

    def extend(self, other):
        base = len(self.calls)
        self.calls.extend(other.calls)
        self.keep.extend(other.keep)
        for i, v in other.flops.items():
            self.flops[base + i] = v

    def run_timed(self, st, pick, overhead_ms=0.0):
        """Replay eagerly with a HIP event pair around every launch `pick(index, name)` selects;
        returns [(index, ms)] (events sit on the launch stream `st`; `overhead_ms`, see
        `event_overhead`, is taken off every reading)."""
        lib = load()
        pairs = []
        for i, (name, fn, args) in enumerate(self.calls):
            if pick(i, name):
                a, b = vp(), vp()
                check(lib.mdm_event_create(C.byref(a))); check(lib.mdm_event_create(C.byref(b)))
                check(lib.mdm_event_record(a, st))
                check(fn(*args, st), name)
                check(lib.mdm_event_record(b, st))
                pairs.append((i, a, b))
            else:
                check(fn(*args, st), name)
        out = []
        for i, a, b in pairs:
            ms = f32()
            check(lib.mdm_event_elapsed_ms(a, b, C.byref(ms)))
            out.append((i, max(ms.value - overhead_ms, 0.0)))
            lib.mdm_event_destroy(a); lib.mdm_event_destroy(b)
        return out

    def event_overhead(self, st, pick, n=24):
        """What an event pair adds to the launch it brackets: for the first `n` picked calls (they must be
        idempotent: forward contractions) compare a pair around ONE launch with a pair around TWO back-to-back
        launches; overhead = 2 T1 - T2.  Median over the sample, in ms."""
        lib = load()
        def timed(fn, args, reps):
            a, b = vp(), vp()
            check(lib.mdm_event_create(C.byref(a))); check(lib.mdm_event_create(C.byref(b)))
            check(lib.mdm_event_record(a, st))
            for _ in range(reps):
                check(fn(*args, st))
            check(lib.mdm_event_record(b, st))
            check(lib.mdm_stream_sync(st))
            ms = f32()
            check(lib.mdm_event_elapsed_ms(a, b, C.byref(ms)))
            lib.mdm_event_destroy(a); lib.mdm_event_destroy(b)
            return ms.value
        est = []
        for i, (name, fn, args) in enumerate(self.calls):
            if len(est) >= n:
                break
            if name == "mdm_gemm" and pick(i, name):
                timed(fn, args, 1)                      # warm
                t1 = min(timed(fn, args, 1) for _ in range(3))
                t2 = min(timed(fn, args, 2) for _ in range(3))
                est.append(2.0 * t1 - t2)
        est.sort()
        return max(est[len(est) // 2], 0.0) if est else 0.0


_recording = None


class GraphExec:
    """A hipGraph instantiated from a Recording (captured on a private stream, replayed on the
    caller's current stream)."""

    def __init__(self, rec):
        self.rec = rec                      # keeps descriptors/tensors alive
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        handle = vp()
        with torch.cuda.stream(side):
            check(load().mdm_graph_begin(side.cuda_stream), "mdm_graph_begin")
            try:
                rec.run(side.cuda_stream)
            finally:
                rc = load().mdm_graph_end(side.cuda_stream, C.byref(handle))
            check(rc, "mdm_graph_end")
        cur.wait_stream(side)
        self.handle = handle

    def launch(self, st=None):
        check(load().mdm_graph_launch(self.handle, stream() if st is None else st), "mdm_graph_launch")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                load().mdm_graph_destroy(self.handle)
        except Exception:
            pass


def call(name, *args):
    """Call a kernel entry point (last positional argument = stream) or record it."""
    fn = getattr(load(), name)
    if _recording is not None:
        _recording.calls.append((name, fn, args[:-1]))
        _recording.keep.append(args)
        return
    check(fn(*args), name)


def ptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    """The HIP stream kernels are launched on: torch's current stream of the current device."""
    if _recording is not None:
        return None        # filled in at replay time
    if not torch.cuda.is_available():
        return None        # no device: argument checks still run, any launch then fails loudly in HIP
    return torch.cuda.current_stream().cuda_stream


def torch_dtype(dt):
    return torch.float32 if dt == F32 else torch.bfloat16


def _desc(kw):
    d = GemmDesc()
    d.alpha = 1.0
    d.batch = 1
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        setattr(d, k, v)
    return d


def gemm_plan(**kw):
    """(split count, workspace bytes) mdm_gemm would use for these fields given unlimited workspace; no launch."""
    kw.pop("_flops", None)
    d = _desc(kw)
    sk, nb = i32(), i64()
    check(load().mdm_gemm_plan(C.byref(d), C.byref(sk), C.byref(nb)), "mdm_gemm_plan")
    return sk.value, nb.value


class WgradGroup:
    """Handle of mdm_wgrad_group_*: a set of weight-gradient descriptors that run as ONE launch (+ one launch summing
    their split-K partials).  `fields_list`: keyword dicts like `gemm()` takes.  The device table lives in a tensor
    owned here; operands are referenced by pointer, so the caller keeps them alive."""

    def __init__(self, fields_list, device):
        lib = load()
        n = len(fields_list)
        self.flops = sum(f.pop("_flops", 0.0) for f in fields_list)
        arr = (GemmDesc * n)()
        for i, f in enumerate(fields_list):
            d = _desc(f)
            C.memmove(C.byref(arr, i * C.sizeof(GemmDesc)), C.byref(d), C.sizeof(GemmDesc))
        need, h = i64(), vp()
        check(lib.mdm_wgrad_group_create(arr, n, None, 0, C.byref(need), C.byref(h)), "mdm_wgrad_group_create")
        self.table = torch.empty(need.value, dtype=torch.uint8, device=device)
        torch.cuda.synchronize(device)
        check(lib.mdm_wgrad_group_create(arr, n, self.table.data_ptr(), need.value, C.byref(need), C.byref(h)), "mdm_wgrad_group_create")
        assert h.value, "wgrad group was not built"
        self.handle, self.n, self.keep = h, n, fields_list

    def launch(self):
        """Launch (or record) the group on the current stream."""
        if _recording is not None:
            _recording.keep.append(self)
            _recording.flops[len(_recording.calls)] = (self.flops, BF16)
        call("mdm_wgrad_group_launch", self.handle, stream())

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                load().mdm_wgrad_group_destroy(self.handle)
        except Exception:
            pass


class Chain:
    """Handle of mdm_chain_*: consecutive small-map convolutions (each an `mdm_gemm` or an `mdm_gemm_pair`) that run as ONE
    persistent launch with per-image hand-offs between the layers instead of launch boundaries (csrc/gemm.hip chain_kernel).
    `phases`: list of tuples of 1 or 2 GemmDesc.  The descriptors are copied into a device table at creation."""

    def __init__(self, phases, device):
        lib = load()
        flat = [d for ph in phases for d in ph]
        arr = (GemmDesc * len(flat))()
        for i, d in enumerate(flat):
            C.memmove(C.byref(arr, i * C.sizeof(GemmDesc)), C.byref(d), C.sizeof(GemmDesc))
        roles = (i32 * len(phases))(*[len(ph) for ph in phases])
        need, h = i64(), vp()
        check(lib.mdm_chain_create(arr, roles, len(phases), None, 0, C.byref(need), C.byref(h)), "mdm_chain_create")
        self.table = torch.empty(need.value, dtype=torch.uint8, device=device)
        torch.cuda.synchronize(device)
        check(lib.mdm_chain_create(arr, roles, len(phases), self.table.data_ptr(), need.value, C.byref(need), C.byref(h)), "mdm_chain_create")
        assert h.value, "chain was not built"
        self.handle, self.n, self.keep = h, len(phases), phases

    def status(self):
        """0 if every in-kernel wait of the launches so far was satisfied (synchronises)."""
        e = C.c_uint32()
        check(load().mdm_chain_status(self.handle, C.byref(e)), "mdm_chain_status")
        return e.value

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                load().mdm_chain_destroy(self.handle)
        except Exception:
            pass


def chain_accepts(da, db=None):
    return bool(load().mdm_chain_accepts(C.byref(da), C.byref(db) if db is not None else None))


def chained(rec, device, min_len=2):
    """A copy of Recording `rec` in which every maximal run of >= `min_len` consecutive chainable launches (mdm_gemm /
    mdm_gemm_pair calls that mdm_chain_accepts) is ONE mdm_chain_launch.  Same kernels' bodies, same results, fewer launches."""
    lib = load()
    out = Recording()
    out.keep = list(rec.keep)
    out.chains = []
    run = []            # [(index, descs)]

    def flush():
        nonlocal run
        if len(run) >= min_len:
            ch = Chain([d for _, d in run], device)
            out.chains.append(ch)
            out.keep.append(ch)
            fl = [rec.flops[i] for i, _ in run if i in rec.flops]
            if fl:
                out.flops[len(out.calls)] = (sum(f for f, _ in fl), fl[0][1])
            out.calls.append(("mdm_chain_launch", lib.mdm_chain_launch, (ch.handle,)))
        else:
            for i, _ in run:
                if i in rec.flops:
                    out.flops[len(out.calls)] = rec.flops[i]
                out.calls.append(rec.calls[i])
        run = []
    for i, (name, fn, args) in enumerate(rec.calls):
        descs = None
        if name == "mdm_gemm":
            d = args[0]._obj
            if chain_accepts(d):
                descs = (d,)
        elif name == "mdm_gemm_pair":
            da, db = args[0]._obj, args[1]._obj
            if chain_accepts(da, db):
                descs = (da, db)
        if descs is None:
            flush()
            if i in rec.flops:
                out.flops[len(out.calls)] = rec.flops[i]
            out.calls.append(rec.calls[i])
        else:
            run.append((i, descs))
    flush()
    return out


def wgrad_group_accepts(**kw):
    kw.pop("_flops", None)
    return bool(load().mdm_wgrad_group_accepts(C.byref(_desc(kw))))


def gemm_pair(kw_a, kw_b):
    """Two independent contractions as ONE call (mdm_gemm_pair): -> (desc_a, desc_b)."""
    fa, fb = kw_a.pop("_flops", None), kw_b.pop("_flops", None)
    da, db = _desc(kw_a), _desc(kw_b)
    if _recording is not None:
        _recording.keep.append((da, kw_a, db, kw_b))
        fl = lambda f, d: f if f is not None else 2.0 * d.M * d.N * d.K * d.batch
        _recording.flops[len(_recording.calls)] = (fl(fa, da) + fl(fb, db), da.dtype)
    call("mdm_gemm_pair", C.byref(da), C.byref(db), stream())
    return da, db


def gemm(**kw):
    """Fill a descriptor from keyword fields (tensors become device pointers) and launch / record it."""
    flops = kw.pop("_flops", None)
    d = _desc(kw)
    if _recording is not None:
        _recording.keep.append((d, kw))
        # algorithmic FLOPs of this launch, keyed by its index in the recording (bench.py roofline)
        _recording.flops[len(_recording.calls)] = (flops if flops is not None else 2.0 * d.M * d.N * d.K * d.batch, d.dtype)
    call("mdm_gemm", C.byref(d), stream())
    return d

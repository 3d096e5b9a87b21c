"""The two-pass entropy-model build of ChannelwisePriorCDFQuantizer.build_entropy_models
(img-compression/quantizer.py:82-150) as ONE stream-ordered device pipeline:

    pass 1   K1t (K1h for sweeps it does not take): solve with raw lengths -> histogram of bit levels [L, C, N+1]   (quantizer.py:96-105)
             (+ all-reduce over ranks)
    lengths  level counts -> -log2 frequencies -> "n + overhead" table [L, C, N+1]              (:105-112, 171-175)
    pass 2   K1 with that table -> rank indices [L, C, B]; K2 -> histogram [L, C, T]            (:119-140)
             (+ all-reduce over ranks); optionally cut into row chunks with K2 of chunk j on a second stream while
             K1 works on chunk j + 1 (measured: no gain, see __init__)
    models   rank counts -> -log2 frequencies [L, C, T]                                          (:141-146)

Nothing synchronises with the host.  When the -log2 step can be tabulated (entropy.neg_log2_lut: every row of a
histogram holds the same number of samples, so -log2(f32(k + n) / f32(total)) is a function of the count k alone and
the table is built once, on the host, with the reference's own NumPy float32 operations) it is a device lookup.  When
it cannot (2^24 samples per row or more, fractional smoothing) the small count tables go through a STREAM-ORDERED host
stage (asynchronous copy into pinned memory, hipLaunchHostFunc, asynchronous copy back): still no synchronisation, and the
whole step still replays from one HIP graph.  The host function is plain C (`NativeHostStage`: the library's
vbq_host_stage_run with NumPy's own float32 log2 inner loop -- no interpreter lock on the runtime's thread) or, where that
loop cannot be had, the reference's NumPy operations in Python (`HostStage`).

`bench.py` times exactly this object; `ChannelwisePriorCDFQuantizer.build_entropy_models` runs it.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

import ctypes as _C
import glob as _glob
import os as _os

from . import entropy as _entropy
from . import ops
from ._lib import VBQError

_HIP = None


def _hip_runtime():
    """The HIP runtime torch already mapped (for hipLaunchHostFunc, which torch does not expose)."""
    global _HIP
    if _HIP is None:
        cands = _glob.glob(_os.path.join(_os.path.dirname(torch.__file__), "lib", "libamdhip64.so*")) + ["libamdhip64.so"]
        for c in cands:
            try:
                _HIP = _C.CDLL(c)
                break
            except OSError:
                continue
        if _HIP is None:
            raise VBQError("libamdhip64.so not found: the stream-ordered host stage needs the HIP runtime")
        _HIP.hipLaunchHostFunc.restype = _C.c_int
        _HIP.hipLaunchHostFunc.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_void_p]
    return _HIP


class HostStage:
    """A host function between two device stages WITHOUT a synchronisation: device tensors -> pinned memory (asynchronous
    copies), `fn(*host_inputs) -> host_outputs` on the runtime's callback thread (hipLaunchHostFunc, ordered on the stream
    like a kernel, capturable into a HIP graph as a host node), pinned memory -> device tensors.  Used for the handful of
    -log2 values per (lambda, channel) that must come from NumPy's own float32 operations (quantizer.py:105-110,141-146)
    when they cannot be tabulated.

    The callback runs Python on the HIP runtime's thread and needs the GIL: a thread that HOLDS the GIL while it blocks on the
    device with a host node pending (a binding that does not release it around hipFree / hipDeviceSynchronize) would
    deadlock.  torch's own synchronising calls release it.  An exception inside `fn` cannot propagate into the runtime: it is
    kept in `error` and raised by `check()` -- callers of `enqueue()` must call `check()` after a synchronisation before they
    trust the outputs (EntropyModelBuild.check does; the quantizer runs it before any table is read)."""
    _CB = _C.CFUNCTYPE(None, _C.c_void_p)

    def __init__(self, inputs, outputs, fn):
        self.inputs, self.outputs, self.fn = list(inputs), list(outputs), fn
        self.h_in = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in self.inputs]
        self.h_out = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in self.outputs]
        self.error = None
        self._cb = self._CB(self._run)                         # kept alive with the object (a graph may replay it)

    def _run(self, _):
        try:
            res = self.fn(*[h.numpy() for h in self.h_in])
            res = res if isinstance(res, (tuple, list)) else (res,)
            for h, r in zip(self.h_out, res):
                np.copyto(h.numpy(), r)
        except BaseException as e:                             # nothing may propagate into the runtime's thread
            self.error = e

    def enqueue(self):
        st = torch.cuda.current_stream(self.inputs[0].device)
        for h, d in zip(self.h_in, self.inputs):
            h.copy_(d, non_blocking=True)
        rc = _hip_runtime().hipLaunchHostFunc(_C.c_void_p(st.cuda_stream), _C.cast(self._cb, _C.c_void_p), None)
        if rc != 0:
            raise VBQError(f"hipLaunchHostFunc failed ({rc})")
        for h, d in zip(self.h_out, self.outputs):
            d.copy_(h, non_blocking=True)

    def check(self):
        if self.error is not None:
            e, self.error = self.error, None
            raise VBQError(f"host stage failed: {type(e).__name__}: {e}") from e


class _PyUFuncObject(_C.Structure):
    """The head of numpy/ufuncobject.h's PyUFuncObject (a public struct, unchanged from NumPy 1.x to 2.x up to `types`)."""
    _fields_ = [("ob_refcnt", _C.c_ssize_t), ("ob_type", _C.c_void_p), ("nin", _C.c_int), ("nout", _C.c_int), ("nargs", _C.c_int),
                ("identity", _C.c_int), ("functions", _C.POINTER(_C.c_void_p)), ("data", _C.POINTER(_C.c_void_p)),
                ("ntypes", _C.c_int), ("reserved1", _C.c_int), ("name", _C.c_char_p), ("types", _C.POINTER(_C.c_char))]


_LOG2_LOOP = None          # (function pointer, data pointer) of np.log2's float32 inner loop; False: not usable here
_NUMPY_TESTED = ("1.21.0", "2.4.0")      # [first, last) NumPy versions whose PyUFuncObject head is the struct above


def _ufunc_struct_is_readable():
    """May `_PyUFuncObject` be laid over `id(np.log2)`?  Every condition is checked BEFORE the first dereference: on another
    object layout (free-threaded CPython: a wider PyObject_HEAD; Py_TRACE_REFS builds; another interpreter; a NumPy outside the
    range this was read against) the `name` / `types` / `functions[i]` reads would be a segmentation fault, which the
    `try / except` around them cannot catch.  Anything unexpected answers False and the callers keep the Python stage."""
    import sys
    import sysconfig
    if sys.implementation.name != "cpython":
        return False
    if sysconfig.get_config_var("Py_GIL_DISABLED") or sysconfig.get_config_var("Py_TRACE_REFS") or hasattr(sys, "getobjects"):
        return False
    if _C.sizeof(_C.c_ssize_t) != 8 or _C.sizeof(_C.c_void_p) != 8:
        return False
    try:
        v = np.lib.NumpyVersion(np.__version__)
        if not (v >= _NUMPY_TESTED[0] and v < _NUMPY_TESTED[1]):
            return False
    except Exception:
        return False
    if type(np.log2) is not np.ufunc:
        return False
    # the object header as this interpreter lays it out: ob_type at offset 8 must be the ufunc type itself
    if _C.c_void_p.from_address(id(np.log2) + 8).value != id(np.ufunc):
        return False
    return object.__basicsize__ == 16 and np.ufunc.__basicsize__ >= _C.sizeof(_PyUFuncObject)


def numpy_log2_f32_loop():
    """NumPy's own float32 log2 as a C function pointer (the first float32 -> float32 inner loop of the np.log2 ufunc object:
    the one NumPy's type resolution picks), so that a host function WITHOUT the interpreter lock can produce the reference's
    numbers -- NumPy's float32 log2 is neither libm's nor correctly rounded.  The ufunc object is only read where
    `_ufunc_struct_is_readable()` vouches for its layout, and the pointer is trusted only after the library's host routine,
    driven by it, has reproduced entropy.neg_log2_freq bit for bit on a random table; None otherwise (the callers then keep the
    Python stage)."""
    global _LOG2_LOOP
    if _LOG2_LOOP is None:
        _LOG2_LOOP = False
        try:
            if not _ufunc_struct_is_readable():
                return None
            u = _PyUFuncObject.from_address(id(np.log2))
            NPY_FLOAT = 11
            if u.nin == 1 and u.nout == 1 and u.nargs == 2 and 0 < u.ntypes < 64 and u.functions and u.types and u.name == b"log2":
                for i in range(u.ntypes):
                    if ord(u.types[2 * i]) == NPY_FLOAT and ord(u.types[2 * i + 1]) == NPY_FLOAT and u.functions[i]:
                        cand = (int(u.functions[i]), int((u.data[i] if u.data else 0) or 0))
                        rng = np.random.default_rng(7)
                        ok = True
                        for K, dt, sm in ((11, np.int64, 1), (2047, np.int32, 1), (257, np.int64, 0.5)):
                            cnt = rng.gamma(0.3, 4e7 / K, (5, K)).astype(dt)
                            got = np.empty((5, K), np.float32)
                            rc = ops._lib.lib().vbq_host_neg_log2_freq_f32(cnt.ctypes.data, int(dt == np.int32), 5, K, float(sm), 0,
                                                                          cand[0], cand[1], got.ctypes.data, None)
                            ok = ok and rc == 0 and np.array_equal(got, _entropy.neg_log2_freq(cnt, sm))
                        if ok:
                            _LOG2_LOOP = cand
                        break
        except Exception:                                       # an interpreter / NumPy build this does not understand
            _LOG2_LOOP = False
    return _LOG2_LOOP or None


class NativeHostStage:
    """HostStage for the -log2 step with NO Python on the runtime's callback thread: the callback is the library's
    vbq_host_stage_run (plain C; NumPy's float32 log2 through its own inner loop, see numpy_log2_f32_loop), so the hazard of
    HostStage -- a thread that holds the interpreter lock while it blocks on the device with the host node pending -- does not
    exist.  counts [..., K] (int32 / int64, device) -> out_model (-log2 of the smoothed frequencies) and, optionally, out_len
    (level + that: quantizer.py:171-175), float32 device tensors of the same shape."""

    def __init__(self, counts: torch.Tensor, out_model: torch.Tensor, out_len: Optional[torch.Tensor], add_n_smoothing, loop):
        self.counts, self.out_model, self.out_len = counts, out_model, out_len
        self.h_in = torch.empty(counts.shape, dtype=counts.dtype, pin_memory=True)
        self.h_model = torch.empty(counts.shape, dtype=torch.float32, pin_memory=True)
        self.h_len = torch.empty(counts.shape, dtype=torch.float32, pin_memory=True) if out_len is not None else None
        K = counts.shape[-1]
        d = self.desc = ops._lib.HostStageDesc()
        d.h_counts, d.counts_are_i32, d.add_level = self.h_in.data_ptr(), int(counts.dtype == torch.int32), int(out_len is not None)
        d.n_rows, d.K, d.add_n_smoothing, d.status, d.runs = counts.numel() // K, K, float(add_n_smoothing), 0, 0
        d.log2_loop, d.log2_data = loop
        d.h_out_model, d.h_out_len = self.h_model.data_ptr(), (self.h_len.data_ptr() if self.h_len is not None else None)
        self._fn = _C.cast(ops._lib.lib().vbq_host_stage_run, _C.c_void_p)
        self.error = None

    def enqueue(self):
        st = torch.cuda.current_stream(self.counts.device)
        self.h_in.copy_(self.counts, non_blocking=True)
        rc = _hip_runtime().hipLaunchHostFunc(_C.c_void_p(st.cuda_stream), self._fn, _C.byref(self.desc))
        if rc != 0:
            raise VBQError(f"hipLaunchHostFunc failed ({rc})")
        self.out_model.copy_(self.h_model, non_blocking=True)
        if self.out_len is not None:
            self.out_len.copy_(self.h_len, non_blocking=True)

    def check(self):
        if self.desc.status != 0:
            rc, self.desc.status = self.desc.status, 0
            raise VBQError(f"host stage failed: vbq_host_neg_log2_freq_f32 returned {rc}")


def chunk_bounds(rows: int, n_chunks: int, unit: int = 2048):
    """Row ranges of the overlap chunks: multiples of `unit` rows (4 workgroups x 512 elements per channel keep the
    persistent K1 grid balanced; any multiple of 8 keeps the vector paths), the remainder in the last chunk."""
    if n_chunks <= 1 or rows < 2 * unit:
        return [(0, rows)]
    units = rows // unit
    n_chunks = min(n_chunks, units)
    base, rem = divmod(units, n_chunks)
    cuts, r = [0], 0
    for j in range(n_chunks):
        r += (base + (1 if j < rem else 0)) * unit
        cuts.append(r)
    cuts[-1] = rows
    return list(zip(cuts[:-1], cuts[1:]))


class EntropyModelBuild:
    """Buffers, tables and streams of the alternation for one problem size; `run()` only enqueues work."""

    def __init__(self, rows: int, n_ch: int, lambdas: Sequence[float], table_lm: torch.Tensor, *, N: int = 10,
                 add_n_smoothing=1, global_rows: Optional[int] = None, distributed: bool = False, group=None,
                 level_group=None, n_chunks: Optional[int] = None, counts_dtype=None, keep_models: bool = True,
                 buffers: Optional[dict] = None, reserved_workgroups: Optional[int] = None):
        """buffers: a dict the caller keeps between builds -- the index planes, the solve's workspace and the -log2 tables
        (device copies) of one shape are taken from it instead of being allocated / computed / uploaded again; every
        OUTPUT tensor (histograms, length table, models) is new per object."""
        self.rows, self.C, self.N = int(rows), int(n_ch), int(N)
        self.lambdas = [float(l) for l in lambdas]
        self.L = len(self.lambdas)
        self.T = ops.table_size(N)
        self.table = table_lm
        self.dev = table_lm.device
        self.smooth = add_n_smoothing
        self.group = group
        # The bit-length histogram's small all-reduce sits on the critical path (pass 2 needs its result); on its own
        # communicator it does not queue behind the previous step's large, asynchronous rank-histogram all-reduce.
        self.level_group = level_group if level_group is not None else group
        # distributed: rows are sharded over the ranks of `group`; the two histograms are summed over them
        self.world = torch.distributed.get_world_size(group) if distributed else 1
        if distributed and self.world > 1 and global_rows is None:
            # the tabulated -log2 (and the packed all-reduce) are only valid for the TRUE number of rows behind a histogram row
            raise ValueError("EntropyModelBuild(distributed=True) needs global_rows = the number of rows summed over all ranks")
        self.global_rows = int(global_rows if global_rows is not None else rows)
        L, C, N1, T = self.L, self.C, N + 1, self.T
        if counts_dtype is None:       # int32 halves the all-reduce payload; exact while no bin can reach 2^31
            counts_dtype = torch.int32 if self.global_rows < 2 ** 31 else torch.int64
        buf = buffers if buffers is not None else {}

        def kept(key, make):
            t = buf.get(key)
            if t is None:
                if len(buf) > 16:             # a caller that walks through many shapes does not pile them up
                    buf.clear()
                t = buf[key] = make()
            return t
        self._kept = kept
        self._idx = None                      # the index planes [L, C, rows], taken (from `buffers`) when a staged pass needs them
        self.level_counts = torch.empty((L, C, N1), dtype=torch.int64, device=self.dev)      # zeroed at the head of every pass 1
        # two rank-histogram buffers when sharded: step i's all-reduce runs while step i+1 fills the other one
        # (pass 2 assigns every bin, or zeroes the buffer itself before it accumulates: no fill here -- 67 MB at C = 256)
        self._counts2 = [torch.empty((L, C, T), dtype=counts_dtype, device=self.dev) for _ in range(2 if self.world > 1 else 1)]
        self._slot = 0
        self.counts = self._counts2[0]
        self.ws = kept(("ws", L, C, N, str(self.dev)),
                       lambda: torch.empty(ops._lib.lib().vbq_quantize_workspace_bytes(C, L, N), dtype=torch.uint8, device=self.dev))

        def lut(K, max_entries):              # (tensor | None,): None is a result worth keeping too
            def make():
                h = _entropy.neg_log2_lut(self.global_rows, K, add_n_smoothing, max_entries=max_entries)
                return (torch.from_numpy(h).to(self.dev) if h is not None else None,)
            return kept(("lut", self.global_rows, K, repr(add_n_smoothing), max_entries, str(self.dev)), make)[0]
        self.lut1 = lut(N1, max(1 << 20, 4 * L * C * N1))
        self.lut2 = lut(T, max(1 << 20, L * C * T)) if keep_models else None
        self.level_len = torch.empty((L, C, N1), dtype=torch.float32, device=self.dev)
        self.raw_models = torch.empty((L, C, N1), dtype=torch.float32, device=self.dev)
        # the model table is worth keeping on the device when it is small or can be tabulated; otherwise (C > 1 without a
        # table: L x C x T values) it is left to the caller (entropy.neg_log2_freq on the counts)
        self.models = torch.empty((L, C, T), dtype=torch.float32, device=self.dev) \
            if keep_models and (self.lut2 is not None or L * C * T <= (1 << 22)) else None
        smooth = add_n_smoothing
        lv = np.arange(N1, dtype=np.int32).astype(np.float32)

        def lengths_host(level_counts):                       # quantizer.py:105-112 and the "n + overhead" of :171-175
            raw = _entropy.neg_log2_freq(level_counts, smooth)
            return (lv + raw).astype(np.float32), raw

        # The -log2 steps that cannot be tabulated: plain C on the runtime's callback thread with NumPy's own log2 loop
        # (NativeHostStage) when this interpreter lets us have it and the smoothing is a Python number (a NumPy float64 scalar
        # would make `c += n` a float64 addition); otherwise -- or with VBQ_PYTHON_HOST_STAGE=1 -- NumPy itself on that thread.
        loop = None
        if type(add_n_smoothing) in (int, float) and T <= 8192 and _os.environ.get("VBQ_PYTHON_HOST_STAGE") != "1":
            loop = numpy_log2_f32_loop()
        self.host_stage_kind = None
        self._len_stage = None
        if self.lut1 is None:
            self._len_stage = NativeHostStage(self.level_counts, self.raw_models, self.level_len, add_n_smoothing, loop) if loop else \
                HostStage([self.level_counts], [self.level_len, self.raw_models], lengths_host)
            self.host_stage_kind = "native" if loop else "python"
        self._model_stages = None
        if self.lut2 is None and self.models is not None:      # quantizer.py:141-146 on the (global) rank counts
            self._model_stages = [NativeHostStage(c, self.models, None, add_n_smoothing, loop) if loop else
                                  HostStage([c], [self.models], lambda counts: _entropy.neg_log2_freq(counts, smooth))
                                  for c in self._counts2]
            self.host_stage_kind = "native" if loop else "python"
        if n_chunks is None:
            # Measured on the Kodak-24 sweep (profiles/r2_overlap_sweep.txt): K2 one chunk behind K1 on a second stream
            # is SLOWER than running them back to back (1.06 / 1.10 / 1.19 ms per step with 2 / 3 / 6 chunks against
            # 0.99 ms) -- K1 saturates the vector ALU, so a co-resident K2 wave only takes issue slots from it.  The
            # chunked form stays available (n_chunks > 1) for experiments; the default is one launch each.
            n_chunks = 1
        self.chunks = chunk_bounds(self.rows, n_chunks)
        self.k1_workgroups_per_cu = 3  # of the 4 that fit: the fourth's LDS and wave slots are K2's while they overlap (chunked form only)
        self.side = torch.cuda.Stream(device=self.dev) if len(self.chunks) > 1 else None
        self._events = [torch.cuda.Event() for _ in self.chunks] if self.side is not None else []
        self.reducers = [None, None]
        self.works = [None, None]
        self._models_pending = [False, False]      # sharded builds: buffers whose model table is still to be looked up
        self.timers = None             # bench.py: callable(name, phase) recording an event on the current stream
        self.collectives = True        # bench.py switches them off to measure what they cost
        if self.world > 1 and counts_dtype == torch.int32:
            from .dist import CountsAllReduce
            self.reducers = [CountsAllReduce(L * C * T, self.dev, max_global_count=self.global_rows, group=group) for _ in range(2)]
        # Launch policy beside the overlapped all-reduce (the `reserved_workgroups` argument of the solve calls; EXPERIMENTS.md, "resident grids beside a
        # collective"): the solve kernels' resident grids assume every workgroup slot of the chip; with a collective's kernel
        # holding some of them K1 takes 1.5 x as long (Kodak-24: 346 -> 530 us with as few as 8 slots taken) against 1.25 x
        # when it leaves those slots alone or runs as short-lived workgroups -- which costs 4-5 % when nothing runs beside it.
        # So: slots are reserved only when the rank histogram's all-reduce is long enough to sit beside the next step's kernels
        # (megabytes: C > 1 builds); the 524 KB of a C = 1 build are over in tens of microseconds and every slot stays K1's.
        if reserved_workgroups is None:
            payload = 0
            if self.world > 1:
                r = self.reducers[0]
                payload = r.payload_bytes(self.counts) if r is not None else self.counts.numel() * self.counts.element_size()
            reserved_workgroups = 64 if payload >= (4 << 20) else 0
        self.reserved_workgroups = int(reserved_workgroups)

    @property
    def idx(self) -> torch.Tensor:
        if self._idx is None:
            L, C = self.L, self.C
            self._idx = self._kept(("idx", L, C, self.rows, str(self.dev)),
                                   lambda: torch.empty((L, C, self.rows), dtype=torch.uint16, device=self.dev))
        return self._idx

    # ---------------------------------------------------------------- the whole build as one C call
    @property
    def one_call_ok(self) -> bool:
        """vbq_build_entropy_models_f32 applies: one GPU, both -log2 steps tabulated (or no model table wanted), no chunking."""
        return (self.world == 1 and self.side is None and self.lut1 is not None and
                (self.models is None or self.lut2 is not None) and self.timers is None)

    def run_one_call(self, means_bc: torch.Tensor, spread_bc: torch.Tensor, spread: str = "sigma"):
        """The whole alternation on channel-last latents [rows, C] in ONE C call (planes, pass 1, length table, pass 2, histogram
        + models): the same launches as `run()` behind one ctypes call.  spread: what spread_bc holds ('sigma' | 'variance' |
        'logvar', see ops.prep_planes)."""
        if not self.one_call_ok:
            raise VBQError("run_one_call: this build needs the staged form (sharded, chunked, timed, or a -log2 step that is not tabulated)")
        means_bc = ops._dev(means_bc, torch.float32, "means")
        spread_bc = ops._dev(spread_bc, torch.float32, "spread")
        if tuple(means_bc.shape) != (self.rows, self.C) or means_bc.shape != spread_bc.shape:
            raise ValueError(f"expected two [{self.rows}, {self.C}] tensors, got {tuple(means_bc.shape)} / {tuple(spread_bc.shape)}")
        h = ops._lib.lib()
        wsb = h.vbq_build_entropy_models_workspace_bytes(self.rows, self.C, self.L, self.N)
        ws = self._kept(("ws_build", self.L, self.C, self.rows, self.N, str(self.dev)),
                        lambda: torch.empty(max(wsb, 256), dtype=torch.uint8, device=self.dev))
        fused = self.models is not None
        ops._lib.check(h.vbq_build_entropy_models_f32(
            ops._ptr(means_bc), ops._ptr(spread_bc), ops._spread_kind(spread), self.rows, self.C, ops._ptr(self.table),
            ops._doubles(self.lambdas), self.L, self.N, ops._ptr(self.lut1), self.lut1.numel(),
            ops._ptr(self.lut2) if fused else None, self.lut2.numel() if fused else 0, ops._ptr(self.level_counts),
            ops._ptr(self.level_len), ops._ptr(self.raw_models), ops._ptr(self.counts), int(self.counts.dtype == torch.int32),
            ops._ptr(self.models) if fused else None, ops._ptr(ws), ws.numel(), self._reserved(), ops._stream(means_bc)),
            "vbq_build_entropy_models_f32")
        self._models_current = fused
        return self

    # ---------------------------------------------------------------- stages
    def _reserved(self) -> int:
        """Workgroup slots THIS build's solve launches leave to a collective's kernel: an argument of every call (the library
        keeps no launch state -- builds with different policies may run side by side, in threads or on several GPUs)."""
        return self.reserved_workgroups if self.collectives else 0

    def pass1(self, mu_cb, sg_cb, level_len=None):
        """quantizer.py:96-105.  level_len: the table of an earlier build, if any (the reference reuses it, :166)."""
        self.level_counts.zero_()
        self._t("k1h", 0)
        ops.level_counts(mu_cb, sg_cb, self.table, self.lambdas, N=self.N, level_len=level_len, layout="cb",
                         out=self.level_counts, workspace=self.ws, reserved_workgroups=self._reserved())
        self._t("k1h", 1)
        if self.world > 1 and self.collectives:
            torch.distributed.all_reduce(self.level_counts, group=self.level_group)
        return self.level_counts

    def _t(self, name, phase):
        if self.timers is not None:
            self.timers(name, phase)

    def lengths(self):
        """quantizer.py:105-112 and the "n + overhead" of :171-175 -> (level_len, raw_models), device f32 [L, C, N+1]."""
        N1 = self.N + 1
        if self.lut1 is not None:
            ops_out = ops._lib.lib().vbq_code_lengths_from_counts
            ops._lib.check(ops_out(ops._ptr(self.level_counts), 0, self.level_counts.numel(), ops._ptr(self.lut1),
                                   self.lut1.numel(), N1, ops._ptr(self.level_len), ops._ptr(self.raw_models),
                                   ops._stream(self.level_counts)), "vbq_code_lengths_from_counts")
        else:                          # the [L, C, N+1] table through a stream-ordered host stage (no synchronisation)
            self._len_stage.enqueue()
        return self.level_len, self.raw_models

    def pass2(self, mu_cb, sg_cb, level_len):
        """quantizer.py:119-140: indices and their per-(lambda, channel) histogram."""
        if self.world > 1:
            self._slot ^= 1
            self.counts = self._counts2[self._slot]
        self.wait(self._slot)                 # the all-reduce that last used this buffer
        main = torch.cuda.current_stream(self.dev)
        self._models_current = False
        if self.side is None and self.world == 1:
            # one GPU: K2 owns whole rows of bins, so it assigns them (no zeroing of the 67 MB array) and, when the code
            # lengths are tabulated, looks them up in the same flush (quantizer.py:141-146)
            fused = self.lut2 is not None and self.models is not None
            self._t("k1", 0)
            ops.quantize(mu_cb, sg_cb, self.table, self.lambdas, N=self.N, level_len=level_len, layout="cb",
                         out_idx=self.idx, workspace=self.ws, reserved_workgroups=self._reserved())
            self._t("k1", 1)
            self._t("k2", 0)
            ops.histogram_models(self.idx, self.C, self.counts, N=self.N, lut=self.lut2 if fused else None,
                                 models=self.models if fused else None)
            self._t("k2", 1)
            self._models_current = fused
            return self.idx, self.counts
        self.counts.zero_()
        if self.side is None:
            self._t("k1", 0)
            ops.quantize(mu_cb, sg_cb, self.table, self.lambdas, N=self.N, level_len=level_len, layout="cb",
                         out_idx=self.idx, workspace=self.ws, reserved_workgroups=self._reserved())
            self._t("k1", 1)
            self._t("k2", 0)
            ops.histogram(self.idx, self.C, N=self.N, layout="cb", out=self.counts)
            self._t("k2", 1)
        else:
            self.side.wait_stream(main)
            for j, (r0, r1) in enumerate(self.chunks):
                self._t("k1", 0)
                ops.quantize(mu_cb, sg_cb, self.table, self.lambdas, N=self.N, level_len=level_len, layout="cb",
                             out_idx=self.idx, workspace=self.ws, rows=(r0, r1), workgroups_per_cu=self.k1_workgroups_per_cu,
                             reserved_workgroups=self._reserved())
                self._t("k1", 1)
                self._events[j].record(main)
                with torch.cuda.stream(self.side):
                    self.side.wait_event(self._events[j])
                    self._t("k2", 0)
                    ops.histogram(self.idx, self.C, N=self.N, layout="cb", out=self.counts, rows=(r0, r1))
                    self._t("k2", 1)
            main.wait_stream(self.side)
        if self.world > 1 and self.collectives:
            if self.reducers[self._slot] is not None:
                self.works[self._slot] = self.reducers[self._slot].start(self.counts)
            else:
                self.works[self._slot] = torch.distributed.all_reduce(self.counts, group=self.group, async_op=True)
        return self.idx, self.counts

    @property
    def reducer(self):
        return self.reducers[self._slot]

    def wait(self, slot=None):
        """The rank histogram's all-reduce (asynchronous, overlapping whatever was enqueued after pass2) is complete
        for the compute stream after this (slot None: both buffers)."""
        for s in ((0, 1) if slot is None else (slot,)):
            if self.works[s] is not None:
                self.works[s].wait()
                self.works[s] = None

    def _models_from(self, slot):
        """counts of buffer `slot` (global once its all-reduce is waited for) -> self.models."""
        self.wait(slot)
        counts = self._counts2[slot]
        if self.lut2 is None:                              # not tabulated: NumPy on the counts, stream-ordered
            self._model_stages[slot].enqueue()
        else:
            ops._lib.check(ops._lib.lib().vbq_code_lengths_from_counts(
                ops._ptr(counts), int(counts.dtype == torch.int32), counts.numel(), ops._ptr(self.lut2),
                self.lut2.numel(), 0, None, ops._ptr(self.models), ops._stream(counts)), "vbq_code_lengths_from_counts")
        self._models_pending[slot] = False

    def finish_models(self):
        """quantizer.py:141-146 -> f32 [L, C, T] on the device (table form) for the MOST RECENT pass 2, or None when the caller
        must take the counts to the host (entropy.neg_log2_freq)."""
        self.wait(self._slot)
        if self.models is None:
            return None
        if getattr(self, "_models_current", False):       # K2 wrote them in its flush
            return self.models
        self._models_from(self._slot)
        self._models_current = True
        return self.models

    @property
    def has_host_stages(self) -> bool:
        """Some -log2 step of this build runs on the HIP runtime's callback thread (`host_stage_kind`: "native" = plain C with
        NumPy's log2 loop; "python" = NumPy in Python, which needs the interpreter lock there)."""
        return self._len_stage is not None or bool(self._model_stages)

    @property
    def graph_safe(self) -> bool:
        """The whole step is stream-ordered work on one stream: it can be captured into a HIP graph."""
        return self.world == 1 and self.side is None

    @property
    def length_table_route(self) -> str:
        return "device (tabulated -log2)" if self.lut1 is not None else \
            f"stream-ordered host stage ({self._stage_text()} on [L, C, N+1] counts, hipLaunchHostFunc; no synchronisation)"

    @property
    def models_route(self) -> str:
        if self.models is None:
            return "not in the step (counts stay on the device)"
        return "device (tabulated -log2)" if self.lut2 is not None else \
            f"stream-ordered host stage ({self._stage_text()} on [L, C, T] counts, hipLaunchHostFunc; no synchronisation)"

    def _stage_text(self) -> str:
        return "plain C with NumPy's float32 log2 loop, no interpreter lock" if self.host_stage_kind == "native" else \
            "NumPy float32 -log2 in Python"

    def check(self):
        """Outside the timed region (synchronises): the assumptions no kernel can see.  (1) The tabulated -log2 is only valid
        when every histogram row holds exactly `global_rows` samples -- a wrong `global_rows` would clamp counts to the
        table's last entry and give silently wrong models.  (2) The packed 3 x 21-bit all-reduce must not have carried
        between fields.  (3) A host stage must not have failed on the callback thread."""
        self.wait()
        for st in [self._len_stage] + list(self._model_stages or []):
            if st is not None:
                torch.cuda.synchronize(self.dev)
                st.check()
        if self.collectives or self.world == 1:
            sums = self.level_counts.sum(dim=-1)
            if not bool(torch.all(sums == self.global_rows)):
                raise VBQError(f"bit-length histogram rows hold {int(sums.min())}..{int(sums.max())} samples, not global_rows = "
                               f"{self.global_rows}: the tabulated code lengths of this build are invalid")
            sums = self.counts.sum(dim=-1, dtype=torch.int64)
            if not bool(torch.all(sums == self.global_rows)):
                raise VBQError(f"rank histogram rows hold {int(sums.min())}..{int(sums.max())} samples, not global_rows = "
                               f"{self.global_rows}: the tabulated models of this build are invalid")
        for r in self.reducers:
            if r is not None:
                r.check()

    def run(self, mu_cb, sg_cb, level_len=None, models: bool = True):
        """One whole alternation; everything is enqueued on the current stream (+ the side stream), nothing waits."""
        self.pass1(mu_cb, sg_cb, level_len)
        ll, _ = self.lengths()
        self.pass2(mu_cb, sg_cb, ll)
        if models and self.models is not None:
            if self.world == 1:
                self.finish_models()
            else:
                # Sharded: this step's rank histogram is being all-reduced asynchronously; the model table of the PREVIOUS
                # step's histogram (the other buffer: its all-reduce ran under this step's kernels) is looked up now, so
                # every step carries one model lookup as on one GPU, one step late.  finish_models() after the last step
                # flushes the pipeline.
                prev = self._slot ^ 1
                if self._models_pending[prev]:
                    self._models_from(prev)
                self._models_pending[self._slot] = True
        return self

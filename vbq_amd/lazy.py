"""Per-image results that stay on the device until somebody reads them on the host.

`ChannelwisePriorCDFQuantizer.compress_latents` / `.compress` (img-compression/quantizer.py:190-256) return, per quantity
('Z_hat', 'raw_num_bits', 'num_bits', 'X_hat') and lambda, an array shaped like the latents.  The reference's arrays are NumPy
(its np.reshape moves everything to the host, :237): 75 MB per Kodak image x 16 lambdas, 2.7 ms over PCIe against 0.09 ms for the
kernels -- although the caller of the evaluation loop (utils.py:542-554) only wants L sums of 'num_bits' and hands 'Z_hat'
straight back to the decoder.  So `return_np=True` returns `LazyArray`s: ndarray-like views (shape, dtype, indexing, arithmetic,
np.* functions, pickling as the ndarray they stand for) over the device tensors the kernels wrote.  Nothing crosses PCIe until
a value is read on the host; consumers that know about the device (`compress` with a torch VAE, `vbq_amd.utils.evaluate_*`)
take `.tensor` and never trigger a copy.

First host access of a quantity copies THAT quantity of the call -- all its lambdas at once -- through a persistent pinned
staging block and out of it into an ordinary array (a pool thread: NumPy releases the GIL for large copies).  A loop reads the same quantities image after image, so the first access in
a call also starts the DMA of those siblings that the PREVIOUS call's results were read for: a caller that reads everything pays
what the eager form paid (the transfers run side by side), one that reads one quantity a third of it, one that reads nothing
nothing.  (Starting the DMA of ALL siblings was measured and dropped: the copies nobody reads sit on the stream -- and the PCIe
link -- in front of the next image's work; 3.6 ms per image for num_bits alone.)  Not thread-safe (one evaluation loop per
quantizer object), like the staging blocks before.

Host copies are read-only arrays (`np.asarray(lazy)[0] = 1` raises, as for any read-only ndarray); `lazy[key] = value` and
`np.add(a, b, out=lazy)` are supported and update the host copy AND the device tensor, so `.tensor`, `compress()`'s decoder input
and the device-side sums never see stale data.  `.copy()` / arithmetic give ordinary writable arrays.
"""
from __future__ import annotations

import itertools
import sys
import weakref
from typing import Dict, Optional

import numpy as np
import torch
from numpy.lib.mixins import NDArrayOperatorsMixin

from .ops import raw_stream as _raw_stream

_THREADS = 3
_THREADED_FROM = 1 << 20           # bytes: below this the calling thread copies (waking the pool costs more)
_ids = itertools.count(1)


class HostStager:
    """The persistent pinned staging blocks of one quantizer (one per quantity name, grown on demand) and the small thread
    pool that copies results out of them into ordinary pageable arrays -- so that a result kept for later does not pin
    page-locked memory (an evaluation loop over a data set would otherwise accumulate GBs of it).

    One asynchronous device -> staging copy per quantity and ONE pool thread per quantity that waits for it and makes the array
    (`np.array` of the staging view: its own allocation, so the three threads fault their pages in separate mappings).  Measured
    (`tools/d2h_bench.py`, 3 x 25 MB): 2.5 ms like this; 3.2-4.4 ms with every quantity cut into pieces for three threads filling
    one array (page faults of one mapping from three threads); 1.4 ms is the DMA alone, 11.5 ms three `tensor.cpu()`."""

    def __init__(self):
        self.blocks: Dict[str, torch.Tensor] = {}
        self.busy: Dict[str, object] = {}        # the copy-out still reading a block (a Future): a new DMA into it waits for that
        self._pool = None
        self._pieces = None
        self.transfers = 0                       # device-to-host copies issued (tests read it)
        self.bytes = 0
        self.spare: Dict[str, list] = {}         # host arrays of results nobody holds any more (DeviceStack.__del__): see land()
        self._group = None                       # (id of) the call whose results are being read, what was read of it ...
        self._reads = set()
        self._prev_reads = set()                 # ... and what was read of the call before: what to prefetch

    def pool(self):
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=_THREADS)
        return self._pool

    def piece_pool(self):
        """Threads that only ever copy (never wait for the device): the pieces of a refilled array."""
        if self._pieces is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pieces = ThreadPoolExecutor(max_workers=2 * _THREADS)
        return self._pieces

    def start(self, stack: "DeviceStack"):
        """Asynchronous device -> staging copy of `stack` and the pool task that turns it into an array (stack._future)."""
        if stack._future is not None or stack._host is not None:
            return
        t = stack.tensor
        prev = self.busy.pop(stack.name, None)
        if prev is not None:
            prev.result()                        # a prefetched sibling of an earlier call nobody has read yet: let its copy finish
            del prev                             # (and let go of its array: the Future holds it as its result)
        h = self.blocks.get(stack.name)
        if h is None or h.numel() < t.numel() or h.dtype != t.dtype:
            h = self.blocks[stack.name] = torch.empty(max(t.numel(), 1), dtype=t.dtype, pin_memory=True)
        st = torch.cuda.current_stream(t.device)
        if stack.stream is not None and stack.stream != st.cuda_stream:
            st.wait_stream(torch.cuda.ExternalStream(stack.stream, device=t.device))      # produced on another stream
            # ... and tell the caching allocator that this stream reads the block: a stack dropped while a prefetched copy is
            # still queued must not have its memory handed out again (on the producing stream) before the copy has run
            t.record_stream(st)
        hv = h[:t.numel()].view(t.shape)
        hv.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(st)
        self.transfers += 1
        self.bytes += t.numel() * t.element_size()

        # The array the caller gets is an ordinary pageable one.  A FRESH 25 MB array costs 1.8-2.2 ms to fill (6 400 page faults;
        # measured: the copy itself is a quarter of that), so arrays of earlier results that nobody references any more are
        # kept (one per quantity) and filled again: an evaluation loop drops image i's results before it reads image i + 1's.
        spare = self.spare.get(stack.name)
        out = None
        while spare:
            cand = spare.pop()
            if cand.shape == tuple(t.shape) and cand.dtype == _NP_DTYPES.get(t.dtype):
                out = cand
                break

        pieces = self.piece_pool() if out is not None and t.dim() and t.shape[0] >= _THREADS else None

        def land():
            ev.synchronize()
            src = hv.numpy()
            if out is None:
                return _frozen(np.array(src))    # a fresh array; the block is free again afterwards
            out.setflags(write=True)             # ours (nobody else holds it): thawed for the refill, frozen again below
            if pieces is None:
                np.copyto(out, src)
                return _frozen(out)
            # pages already there: the copy is plain memory bandwidth, and three threads have more of it than one
            n0 = src.shape[0]
            cuts = [(n0 * j) // _THREADS for j in range(_THREADS + 1)]
            jobs = [pieces.submit(np.copyto, out[a:b], src[a:b]) for a, b in zip(cuts[1:-1], cuts[2:])]
            np.copyto(out[:cuts[1]], src[:cuts[1]])
            for j in jobs:
                j.result()
            return _frozen(out)
        if t.numel() * t.element_size() >= _THREADED_FROM:
            stack._future = self.busy[stack.name] = self.pool().submit(land)
        else:
            stack._future = _Done(land())

    def fetch(self, stack: "DeviceStack") -> np.ndarray:
        """The host copy of `stack`; on the way the transfers of those siblings the PREVIOUS call's results were read for."""
        if self._group != stack.group:           # the first read of another call's results: remember what the last one was read for
            if self._group is not None:
                self._prev_reads = self._reads
            self._group, self._reads = stack.group, set()
        self._reads.add(stack.name)
        self.start(stack)
        for ref in stack.siblings:
            sib = ref()
            if sib is not None and sib is not stack and sib.name in self._prev_reads:
                self.start(sib)
        fut, stack._future = stack._future, None
        out = fut.result()
        if self.busy.get(stack.name) is fut:     # the copy-out is over: the block is free, and the Future (which holds `out` as
            del self.busy[stack.name]            # its result) must not outlive this call -- DeviceStack.__del__ hands an array
        del fut                                  # back for refilling only when nobody else references it
        return out


def _frozen(a: np.ndarray) -> np.ndarray:
    """Host copies are READ-ONLY: the device tensor stays the other consumers' source (`.tensor`, the decoder, the device-side
    sums), and a silent host-only edit would leave the two different.  Writes go through LazyArray.__setitem__ / `out=`, which
    update both."""
    a.setflags(write=False)
    return a


class _Done:
    def __init__(self, value):
        self.value = value

    def result(self):
        return self.value


class DeviceStack:
    """One quantity of one call: a device tensor [L, *shape] and, once somebody has read it, its host copy."""

    def __init__(self, name: str, tensor: torch.Tensor, stager: Optional[HostStager]):
        self.name, self.tensor, self.stager = name, tensor, stager
        self.id = next(_ids)
        self.group = self.id                     # the call it belongs to (see group()); siblings: weak references, no cycles --
        self.siblings = ()                       # a dropped result frees its device tensors and host arrays at once
        self._host = None
        self._future = None                      # a transfer under way (HostStager.start)
        # the stream the producing kernels were enqueued on (its raw handle): a copy issued from another stream waits for it
        self.stream = _raw_stream(tensor.device) if tensor.is_cuda else None

    @property
    def on_device(self) -> bool:
        """True while no host copy has been made."""
        return self._host is None

    def host(self) -> np.ndarray:
        if self._host is None:
            if not self.tensor.is_cuda or self.stager is None:
                self._host = _frozen(self.tensor.detach().cpu().numpy().copy())
            else:
                self._host = self.stager.fetch(self)
        return self._host

    def write(self, i, key, value):
        """row i, [key] = value on BOTH copies: the host array (thawed for the assignment) and the device tensor (the row is
        uploaded again: one small synchronous copy -- writes into results are rare, stale device data would be silent)."""
        h = self.host()
        h.setflags(write=True)
        try:
            h[i][key] = value
        finally:
            h.setflags(write=False)
        self.tensor[i].copy_(torch.from_numpy(np.array(h[i])))              # (np.array: a writable copy, what from_numpy wants)

    def rows(self):
        return [LazyArray(self, i) for i in range(self.tensor.shape[0])]

    def __del__(self):
        # Hand the host array back for the next call's results -- only when nobody else can see it: every view a caller took
        # (np.asarray(lazy) is one) holds a reference to it, so a count of exactly two (the attribute and getrefcount's own
        # argument) means this object was the last owner.
        try:
            st = self.stager
            if (self._host is not None and st is not None and self._host.nbytes >= _THREADED_FROM and self._host.flags.owndata
                    and sys.getrefcount(self._host) == 2):
                keep = st.spare.setdefault(self.name, [])
                if len(keep) < 1:
                    keep.append(self._host)
        except Exception:                        # interpreter shutdown
            pass


def group(stacks):
    """Quantities of ONE call: the first host access of any of them starts the transfers of the others as well."""
    stacks = [s for s in stacks if s is not None]
    refs = tuple(weakref.ref(s) for s in stacks)
    for s in stacks:
        s.siblings, s.group = refs, stacks[0].id
    return stacks


def join(member, new: DeviceStack) -> DeviceStack:
    """Add `new` to the call `member` (a DeviceStack or one of its LazyArrays) belongs to: 'X_hat' of compress() is a quantity of
    the same call as the latents it was decoded from -- on its own it would look like ANOTHER call to the prefetch rule, and a
    loop that reads 'X_hat' and 'num_bits' would never see its siblings' transfers started side by side."""
    st = member._stack if isinstance(member, LazyArray) else member
    mates = [r() for r in st.siblings] if st.siblings else [st]
    group([m for m in mates if m is not None] + [new])
    return new


class LazyArray(NDArrayOperatorsMixin):
    """Row `i` of a DeviceStack, behaving like the NumPy array it stands for.  `.tensor`: the device tensor (no copy);
    everything else goes through `__array__`, which makes the stack's host copy on first use."""
    __slots__ = ("_stack", "_i")
    __array_priority__ = 100.0

    def __init__(self, stack: DeviceStack, i: int):
        self._stack, self._i = stack, i

    # ---- what is known without touching the data
    @property
    def tensor(self) -> torch.Tensor:
        return self._stack.tensor[self._i]

    @property
    def on_device(self) -> bool:
        return self._stack.on_device

    # zero-copy hand-over to device libraries: torch.as_tensor(x, device="cuda"), torch.from_dlpack(x), CuPy, Numba
    @property
    def __cuda_array_interface__(self):
        t = self.tensor
        if not t.is_cuda:
            raise AttributeError("__cuda_array_interface__")
        return t.__cuda_array_interface__

    def __dlpack__(self, *args, **kwargs):
        return self.tensor.__dlpack__(*args, **kwargs)

    def __dlpack_device__(self):
        return self.tensor.__dlpack_device__()

    @property
    def shape(self):
        return tuple(self._stack.tensor.shape[1:])

    @property
    def ndim(self):
        return self._stack.tensor.dim() - 1

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def dtype(self):
        return _NP_DTYPES[self._stack.tensor.dtype]

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of unsized object")
        return self.shape[0]

    # ---- the data
    def __array__(self, dtype=None, copy=None):
        a = self._stack.host()[self._i, ...]             # (the Ellipsis keeps a 0-d row an array, a view as well)
        if dtype is not None and np.dtype(dtype) != a.dtype:
            if copy is False:                    # NumPy 2's protocol: a conversion is a copy
                raise ValueError("a copy is needed to convert the dtype, and copy=False was asked for")
            return a.astype(dtype)
        return a.copy() if copy else a

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        conv = lambda x: np.asarray(x) if isinstance(x, LazyArray) else x
        inputs = tuple(conv(x) for x in inputs)
        outs = kwargs.get("out")
        if outs is None or not any(isinstance(o, LazyArray) for o in outs):
            return getattr(ufunc, method)(*inputs, **kwargs)
        # out= into a result: computed into ordinary arrays, then written to both copies (see DeviceStack.write)
        kwargs["out"] = tuple(np.array(o) if isinstance(o, LazyArray) else o for o in outs)
        res = getattr(ufunc, method)(*inputs, **kwargs)
        for o, tmp in zip(outs, kwargs["out"]):
            if isinstance(o, LazyArray):
                o._stack.write(o._i, Ellipsis, tmp)
        if isinstance(res, tuple):
            return tuple(o if isinstance(o, LazyArray) else r for o, r in zip(outs, res))
        return outs[0] if isinstance(outs[0], LazyArray) else res

    def __getitem__(self, key):
        return np.asarray(self)[key]

    def __setitem__(self, key, value):
        self._stack.write(self._i, key, value)

    def __iter__(self):
        return iter(np.asarray(self))

    def __getattr__(self, name):                 # .sum(), .reshape(), .astype(), .T, ...: the ndarray's own
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(np.asarray(self), name)

    def __bool__(self):
        return bool(np.asarray(self))

    def __float__(self):
        return float(np.asarray(self))

    def __int__(self):
        return int(np.asarray(self))

    def __index__(self):
        return np.asarray(self).__index__()

    def __reduce__(self):                        # pickles as the ndarray it stands for
        return np.asarray(self).__reduce__()

    def __copy__(self):
        return np.asarray(self).copy()

    def __deepcopy__(self, memo):
        return np.asarray(self).copy()

    def __repr__(self):
        where = "on the device" if self.on_device else "host copy made"
        return f"LazyArray(shape={self.shape}, dtype={self.dtype}, {where})"


_NP_DTYPES = {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64), torch.int32: np.dtype(np.int32),
              torch.int64: np.dtype(np.int64), torch.uint16: np.dtype(np.uint16), torch.uint8: np.dtype(np.uint8),
              torch.int16: np.dtype(np.int16), torch.float16: np.dtype(np.float16), torch.bool: np.dtype(np.bool_)}


def device_tensor(x) -> Optional[torch.Tensor]:
    """The device tensor behind `x` when it is a LazyArray (or already a device tensor), else None."""
    if isinstance(x, LazyArray):
        return x.tensor
    if isinstance(x, torch.Tensor) and x.is_cuda:
        return x
    return None


def common_stack(values) -> Optional[torch.Tensor]:
    """When `values` are the LazyArrays 0, 1, 2, ... of ONE DeviceStack, in order: that stack's device tensor [L, ...] (a
    consumer can then work on all lambdas at once); else None."""
    values = list(values)
    if not values or type(values[0]) is not LazyArray:
        return None
    st = values[0]._stack
    for v in values:
        if type(v) is not LazyArray or v._stack is not st:
            return None
    idx = [v._i for v in values]
    if idx == list(range(st.tensor.shape[0])):
        return st.tensor
    return st.tensor[idx]

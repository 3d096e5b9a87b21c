"""One image of the evaluation loop as ONE HIP graph replay.

`utils.evaluate_compression_quantizer` (img-compression/utils.py:542-556) does, image after image,

    tmp = quantizer.compress(X, vae, settings, clip=True)          # quantizer.py:242-256
    np.sum(tmp['num_bits'][s]), np.sum(tmp['num_bits_cl'][s]), uint8(tmp['X_hat'][s])     for every setting s

On the device that is a dozen small launches (three for `compress_latents`, the decoder's, the clamp, two NumPy-order row sums,
the uint8 conversion, three small copies to the host) whose kernels take ~0.1 ms together: the loop is bound by the host's
launch path, not by the device.  All images of one shape run the SAME launches on buffers of the SAME sizes, so they are
captured once -- `compress` itself, the very code path, with the decoder, plus the loop's reads -- into a HIP graph held by the
quantizer (`ChannelwisePriorCDFQuantizer.compress_replay`), and every further image of that shape is one copy of the input into
the graph's static buffer, one `hipGraphLaunch`, one synchronisation.  Kodak has two shapes (landscape, portrait).

What a replay returns is `compress`'s result dict over the graph's STATIC tensors: valid until the next replay of the same shape
(the evaluation loop has consumed its sums and images by then; a caller that keeps per-image latents takes `compress`).

Capture needs the VAE's `encode` / `decode` to be torch code that runs on the device without synchronising.  Tried in this order,
each inside try / except: (1) `encode` inside the graph on the static device copy of X; (2) `encode` outside (as the reference calls
it, NumPy X), its outputs copied into static tensors, everything after it inside; (3) no graph: `compress` + `evaluation_reads` as
before.  The choice is made once per (shape, settings, vae) and reported by `mode`.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch


def _abort_capture(dev):
    """After a capture that failed half way (a decoder that synchronises, an allocation the runtime refuses while capturing): the
    capture stream may be left in capture mode -- and under the global capture mode every later synchronising call of the process
    would fail.  End whatever capture is still open on torch's capture stream, drop the partial graph, clear the sticky error."""
    import ctypes as C
    from .pipeline import _hip_runtime
    try:
        hip = _hip_runtime()
        st = getattr(torch.cuda.graph, "default_capture_stream", None)
        if st is not None:
            status = C.c_int(0)
            hip.hipStreamIsCapturing(C.c_void_p(st.cuda_stream), C.byref(status))
            if status.value != 0:
                g = C.c_void_p()
                hip.hipStreamEndCapture(C.c_void_p(st.cuda_stream), C.byref(g))
                if g.value:
                    hip.hipGraphDestroy(g)
        hip.hipGetLastError()
    except Exception:
        pass
    try:
        torch.cuda.synchronize(dev)
    except Exception:
        pass


class _EncodeProxy:
    """The VAE `compress` sees while it is being captured / replayed: `encode` hands out what the replay prepared."""

    def __init__(self, vae):
        self._vae = vae
        self.latents = None                      # (means, logvars) static tensors (mode "latents") or None (mode "full")
        self.x_dev = None

    def encode(self, X):
        if self.latents is not None:
            return self.latents
        return self._vae.encode(self.x_dev)

    def decode(self, Z):
        return self._vae.decode(Z)


class CompressReplay:
    """`compress(X, vae, lambs, clip)` + the evaluation loop's reads for ONE input shape, captured once, replayed per image."""

    def __init__(self, quantizer, vae, X: np.ndarray, lambs: Sequence, clip: bool):
        self.q, self.vae, self.lambs, self.clip = quantizer, vae, list(lambs), bool(clip)
        self.shape, self.dtype = tuple(X.shape), X.dtype
        self.mode = "eager"
        self.errors: Dict[str, str] = {}         # why a capture form was not taken
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.replays = 0
        self._proxy = _EncodeProxy(vae)
        self._out = None
        self._dev_reads = None
        self._host_reads = None
        if not torch.cuda.is_available() or quantizer.device.type != "cuda":
            return
        dev = quantizer.device
        self._x_host = torch.empty(self.shape, dtype=torch.from_numpy(np.empty(0, self.dtype)).dtype, pin_memory=True)
        self._x_np = self._x_host.numpy()
        self._x_dev = torch.empty(self.shape, dtype=self._x_host.dtype, device=dev)
        for mode in ("full", "latents"):
            try:
                self._capture(mode, X)
                self.mode = mode
                break
            except Exception as e:               # an encoder / decoder that cannot run inside a capture: next form
                self.graph = None
                self.errors[mode] = f"{type(e).__name__}: {e}"[:300]
                _abort_capture(dev)
                continue

    # ------------------------------------------------------------------ capture
    def _prepare(self, mode: str, X: np.ndarray, upload: bool = True):
        """X into the page-locked staging array; in mode "full" the copy to the device is a node of the graph itself (upload=False
        on replays), in mode "latents" the encoder runs here, on the NumPy image, and its outputs go into the static tensors."""
        np.copyto(self._x_np, X)
        if mode == "full":
            if upload:
                self._x_dev.copy_(self._x_host, non_blocking=True)
            self._proxy.latents, self._proxy.x_dev = None, self._x_dev
        else:
            m, lv = self.vae.encode(X)                                            # as the reference calls it
            m = m if isinstance(m, torch.Tensor) else torch.from_numpy(np.asarray(m))
            lv = lv if isinstance(lv, torch.Tensor) else torch.from_numpy(np.asarray(lv))
            if self._proxy.latents is None:
                self._proxy.latents = (torch.empty(m.shape, dtype=torch.float32, device=self.q.device),
                                       torch.empty(lv.shape, dtype=torch.float32, device=self.q.device))
            self._proxy.latents[0].copy_(m, non_blocking=True)
            self._proxy.latents[1].copy_(lv, non_blocking=True)

    def _step(self):
        from . import utils
        out = self.q.compress(self._x_dev if self.mode_being_captured == "full" else self._np_stub, self._proxy, self.lambs, clip=self.clip)
        reads = utils.evaluation_device_reads(out, self.lambs)
        if reads is None:
            raise RuntimeError("results are not device-resident stacks of one call")
        return out, reads

    def _capture(self, mode: str, X: np.ndarray):
        dev = self.q.device
        self.mode_being_captured = mode
        self._np_stub = np.empty(self.shape, self.dtype)                          # compress only takes np.shape(X) of it
        self._proxy.latents = None
        self._prepare(mode, X)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                                            # warm-up outside the capture (allocations, lazy inits)
            for _ in range(2):
                _, reads = self._step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # page-locked memory cannot be allocated while a stream is capturing: the landing buffers of the reads come first
        host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in reads]
        del reads
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            if mode == "full":
                self._x_dev.copy_(self._x_host, non_blocking=True)               # the image's way to the device: a node as well
            out, reads = self._step()
            for h, t in zip(host, reads):                                        # the copies to the host are nodes of the graph too
                h.copy_(t, non_blocking=True)
        self.graph, self._out, self._dev_reads, self._host_reads = g, out, reads, host
        from .lazy import LazyArray
        stacks = {}
        for d in out.values():                                                   # the stacks behind the result dict, once
            for v in d.values():
                if isinstance(v, LazyArray):
                    stacks[id(v._stack)] = v._stack
        self._stacks = list(stacks.values())

    # ------------------------------------------------------------------ replay
    def run(self, X) -> Tuple[Dict, Tuple[np.ndarray, np.ndarray, np.ndarray]]:
        """-> (compress's result dict over the graph's static tensors, (sums of num_bits, sums of num_bits_cl, uint8
        reconstructions) on the host).  Valid until the next run()."""
        from . import utils
        X = np.asarray(X)
        if self.graph is None:
            out = self.q.compress(X, self.vae, self.lambs, clip=self.clip)
            return out, utils.evaluation_reads(out, self.lambs, self.__dict__.setdefault("_staging", {}))
        self._prepare(self.mode, X, upload=False)
        self.graph.replay()
        self.replays += 1
        self._fresh_views()
        torch.cuda.current_stream(self.q.device).synchronize()
        return self._out, tuple(np.array(h.numpy()) for h in self._host_reads)

    def _fresh_views(self):
        """The lazy views of the result dict cache a host copy once read: a replay has new values behind the same tensors."""
        for st in self._stacks:
            st._host, st._future = None, None

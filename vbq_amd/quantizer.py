"""ChannelwisePriorCDFQuantizer with the reference's call surface
(img-compression/quantizer.py:13-256), running on the HIP kernels.

Same constructor, method names, argument meaning, returned dict layout and state attributes
as the reference class, so `post_process.build_cdf_qzer` / `utils.evaluate_compression_quantizer`
(post_process.py:99-107, utils.py:542-554) can use it unchanged.  What differs is inside:

* the whole of get_all_N_bit_intervals -> candidate assembly -> batch_quantize_indep_dims ->
  qidx lookup (quantizer.py:65-80, 156-188, 135/223) is ONE kernel launch (K1) for all lambdas;
* per-channel bincounts (quantizer.py:104-105, 138-140) are one histogram launch (K2);
* latents live on the GPU as channel-major planes [C, B]: every workgroup then works on one
  channel -> one 8 KB code-point table in LDS and wave-uniform penalties.

Arrays may be NumPy arrays or torch tensors; results are NumPy arrays (as in the reference,
whose np.reshape moves everything to the host, quantizer.py:237) unless `return_np=False`.
There is no CPU path: without the HIP extension and a ROCm device the methods raise.
"""
from __future__ import annotations

from collections.abc import MutableMapping
from typing import Dict, Optional

import numpy as np
import torch

from . import entropy as _entropy
from . import ops
from ._lib import VBQError


def n_bit_binary_floats(n: int):
    """utils.py:23-24."""
    return [i * 2 ** (-n) + 2 ** (-n - 1) for i in range(2 ** n)]


def _to_numpy(a):
    if isinstance(a, torch.Tensor):
        return a.detach().cpu().numpy()
    if hasattr(a, "numpy") and not isinstance(a, np.ndarray):      # e.g. a TF eager tensor
        return np.asarray(a.numpy())
    return np.asarray(a)


def _default_device():
    if not torch.cuda.is_available():
        raise VBQError("no ROCm device visible: the VBQ quantizer has no CPU implementation")
    return torch.device("cuda", torch.cuda.current_device())


class DeviceModels(MutableMapping):
    """dict[lamb] -> NumPy array [C, K] whose rows live ON THE DEVICE until somebody reads them on the host.

    build_entropy_models leaves its three tables -- entropy_models, raw_code_length_entropy_models and the integer
    histograms behind them -- where the kernels wrote them: compress_latents reads them there, and the 67 MB of models plus
    the counts (Kodak-24, C = 256, 32 lambdas) cross PCIe only if a caller actually looks at them (`models[lamb]`, iteration
    over values / items, pickling, `save`).  The first host read copies the whole stack once and runs the build's deferred
    checks (EntropyModelBuild.check: histogram totals, host stages); the arrays are read-only, as the eager ones were (an
    in-place edit would leave the device copy stale).  Assigning `models[lamb] = array` works as on a dict; the device
    stack is then no longer used for the lookups (`device_rows` returns None and the caller rebuilds its copy)."""

    def __init__(self, keys, stack: torch.Tensor, companions=None, on_read=None):
        self._keys = list(keys)
        self._index = {k: i for i, k in enumerate(self._keys)}
        self._stack = stack
        self._companions = dict(companions or {})
        self._on_read = on_read
        self._host = None
        self._over = {}                      # entries the caller replaced or added
        self._gone = set()
        self._subsets = {}                   # device rows of key subsets already gathered (an evaluation loop asks again and again)

    @property
    def on_device(self) -> bool:
        """True while no host copy has been made."""
        return self._host is None

    def _materialise(self):
        if self._host is None:
            if self._on_read is not None:
                self._on_read()
            h = self._stack.cpu().numpy()
            h.setflags(write=False)
            self._host = h
            # ONE view object per key, made once: callers that key caches on the arrays' identities (_keyed_dev) see the same
            # object on every read instead of a fresh view per __getitem__ (which re-uploaded the whole table per image)
            self._rows = [h[i] for i in range(h.shape[0])]
        return self._host

    def __getitem__(self, k):
        if k in self._over:
            return self._over[k]
        if k in self._gone:
            raise KeyError(k)
        i = self._index[k]                   # KeyError as a dict
        self._materialise()
        return self._rows[i]

    def __setitem__(self, k, v):
        self._over[k] = v
        self._gone.discard(k)
        self._subsets.clear()

    def __delitem__(self, k):
        self._subsets.clear()
        if k in self._over:
            del self._over[k]
            if k in self._index:
                self._gone.add(k)
        elif k in self._index and k not in self._gone:
            self._gone.add(k)
        else:
            raise KeyError(k)

    def __iter__(self):
        for k in self._keys:
            if k not in self._gone:
                yield k
        for k in self._over:
            if k not in self._index:
                yield k

    def __len__(self):
        return len(self._keys) - len(self._gone) + sum(1 for k in self._over if k not in self._index)

    def __contains__(self, k):
        return k in self._over or (k in self._index and k not in self._gone)

    def device_rows(self, keys, companion=None):
        """The rows of `keys` as a device tensor [len(keys), C, K] (of the companion stack when named), or None when an
        entry was replaced by the caller (the device stack no longer tells the whole story).  KeyError for unknown keys."""
        keys = list(keys)
        for k in keys:
            if k not in self:
                raise KeyError(k)
        if any(k in self._over for k in keys):
            return None
        t = self._stack if companion is None else self._companions[companion]
        if keys == self._keys:
            return t
        which = (companion, tuple(self._index[k] for k in keys))
        sub = self._subsets.get(which)
        if sub is None:
            if len(self._subsets) > 8:
                self._subsets.clear()
            sub = self._subsets[which] = t[list(which[1])].contiguous()
        return sub

    def to_dict(self):
        return {k: self[k] for k in self}

    def __reduce__(self):                    # pickles as the plain dict of NumPy arrays it stands for
        return (dict, (self.to_dict(),))

    def __repr__(self):
        return f"DeviceModels({len(self)} lambdas, {'device-resident' if self.on_device else 'host copy made'})"


class ChannelwisePriorCDFQuantizer:
    def __init__(self, num_channels, max_bits_per_coord, float_type="float32", int_type="int32", device=None,
                 validate_inputs=False):
        if float_type != "float32":
            raise ValueError("only float_type='float32' is supported (the reference default, quantizer.py:14)")
        self.max_bits_per_coord = int(max_bits_per_coord)
        self.num_channels = int(num_channels)
        self.float_type = float_type
        self.int_type = int_type
        self.quantization_levels = 2 ** (self.max_bits_per_coord + 1) - 1          # quantizer.py:20
        self.raw_code_length_entropy_models = None
        self.entropy_models = None
        self.process_group = None          # set to a torch.distributed group to all-reduce the histograms
        self._device = device
        # True: every batch is checked on the device for NaN / infinite means and non-positive / non-finite standard
        # deviations before it is solved (ValueError; costs one synchronisation per call).  The reference has no such
        # check -- NaNs flow through tf.argmax there -- so it is off by default.
        self.validate_inputs = bool(validate_inputs)
        self._dev_cache: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ state / pickling
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_dev_cache"] = {}
        st["_device"] = None
        st["process_group"] = None
        for k, v in list(st.items()):          # device-resident tables travel as the dicts of NumPy arrays they stand for
            if isinstance(v, DeviceModels):
                st[k] = v.to_dict()
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._dev_cache = {}

    # ------------------------------------------------------------------ on-disk format
    def save(self, path):
        """Portable alternative to the reference's pickle (post_process.py:106-107): one .npz holding the
        code-point table, the two entropy-model dicts (keys = lambdas as float64) and the histograms."""
        if not hasattr(self, "all_code_points"):
            raise ValueError("nothing to save: build_code_points() first")
        d = dict(format=np.array("vbq_amd.quantizer/1"), num_channels=self.num_channels,
                 max_bits_per_coord=self.max_bits_per_coord, all_code_points=self.all_code_points)
        if self.entropy_models:
            lambs = self.lambs
            d["lambdas"] = np.array([float(l) for l in lambs], dtype=np.float64)
            d["entropy_models"] = np.stack([self.entropy_models[l] for l in lambs])
            if self.raw_code_length_entropy_models:
                d["raw_code_length_entropy_models"] = np.stack([self.raw_code_length_entropy_models[l] for l in lambs])
            if hasattr(self, "_code_counts"):
                d["code_counts"] = np.stack([self._code_counts[l] for l in lambs])
                d["add_n_smoothing"] = np.array(self._add_n_smoothing)
        np.savez_compressed(path, **d)

    @classmethod
    def load(cls, path, device=None):
        z = np.load(path, allow_pickle=False)
        if str(z["format"]) != "vbq_amd.quantizer/1":
            raise ValueError(f"{path}: unknown format {z['format']}")
        q = cls(int(z["num_channels"]), int(z["max_bits_per_coord"]), device=device)

        class _Table:
            def __init__(self, pts):
                self.pts = pts

            def inverse_cdf(self, xi):
                return self.pts
        q.build_code_points(_Table(np.ascontiguousarray(z["all_code_points"].T)))
        if "lambdas" in z:
            lambs = [float(v) for v in z["lambdas"]]
            q.entropy_models = {l: z["entropy_models"][i] for i, l in enumerate(lambs)}
            if "raw_code_length_entropy_models" in z:
                q.raw_code_length_entropy_models = {l: z["raw_code_length_entropy_models"][i] for i, l in enumerate(lambs)}
            if "code_counts" in z:
                q._code_counts = {l: z["code_counts"][i] for i, l in enumerate(lambs)}
                q._add_n_smoothing = z["add_n_smoothing"].item()
        return q

    @property
    def device(self):
        if self._device is None:
            self._device = _default_device()
        return self._device

    def _dev(self, name: str, builder):
        t = self._dev_cache.get(name)
        if t is None or t.device != self.device:
            t = builder().to(self.device)
            self._dev_cache[name] = t
        return t

    # ------------------------------------------------------------------ tables (quantizer.py:25-63)
    def build_code_points(self, prior_model, **kwargs):
        N, C = self.max_bits_per_coord, self.num_channels
        all_bin_floats = np.hstack([n_bit_binary_floats(n) for n in range(N + 1)])
        all_bin_floats_rep = np.repeat(all_bin_floats[:, None], C, axis=1)          # T x C
        pts = _to_numpy(prior_model.inverse_cdf(all_bin_floats_rep, **kwargs))
        if pts.shape != (self.quantization_levels, C):
            raise ValueError(f"prior.inverse_cdf returned shape {pts.shape}, expected {(self.quantization_levels, C)}")
        self.all_code_points = np.ascontiguousarray(pts.astype(np.float32).T)       # C x T, level-major
        self.code_points_by_channel = np.sort(self.all_code_points, axis=1)         # MUST BE SORTED (:37)
        self.code_points_by_bits = [[self.all_code_points[c, 2 ** n - 1: 2 ** (n + 1) - 1] for n in range(N + 1)]
                                    for c in range(C)]
        grids = np.empty((C, N + 1, 2 ** N), dtype=np.float32)
        for c in range(C):
            for n in range(N + 1):
                lvl = self.code_points_by_bits[c][n]
                pad = 2 ** (N - 1) - 1 if n == 0 else 2 ** (N - 1) - 2 ** (n - 1)
                grids[c, n] = np.pad(np.array([lvl[0]] * 2) if n == 0 else lvl, (pad,), "edge")
        self._search_grids = grids
        # The kernels need the merged table to be non-decreasing in xi (then rank order == sorted order).
        from .tables import rank_of_slot
        in_rank_order = np.empty_like(self.all_code_points)
        in_rank_order[:, rank_of_slot(N)] = self.all_code_points
        if not np.all(np.diff(in_rank_order, axis=1) >= 0):
            raise ValueError("prior.inverse_cdf is not monotone in xi after the float32 cast; the reference's "
                             "per-level searchsorted (quantizer.py:74) is undefined on such tables")
        self._strict = bool(np.all(np.diff(in_rank_order, axis=1) > 0))
        # canonical index of quantizer.py:135/223: first sorted position holding the same value
        self._canon = (np.stack([np.searchsorted(r, r, side="left") for r in in_rank_order]).astype(np.int64)
                       if not self._strict else None)
        self._dev_cache = {}

    def _table_dev(self):
        return self._dev("table_lm", lambda: torch.from_numpy(self.all_code_points))

    def _sorted_dev(self):
        return self._dev("sorted", lambda: torch.from_numpy(self.code_points_by_channel))

    # ------------------------------------------------------------------ compat helper (quantizer.py:65-80)
    def get_all_N_bit_intervals(self, Z):
        """Left/right n-bit neighbours, C x (N+1) x B (device tensors).  Kept for API compatibility only: the
        solve below never materialises these tensors (vbq_n_bit_intervals_f32)."""
        from . import _lib
        Zt = torch.as_tensor(_to_numpy(Z) if not isinstance(Z, torch.Tensor) else Z, dtype=torch.float32).to(self.device)
        if Zt.dim() != 2 or Zt.shape[1] != self.num_channels:
            raise ValueError(f"expected Z of shape [B, {self.num_channels}], got {tuple(Zt.shape)}")
        z_cb = ops.transpose(Zt.contiguous())                                       # [C, B] (quantizer.py:73)
        C, B, N = self.num_channels, Zt.shape[0], self.max_bits_per_coord
        left = torch.empty((C, N + 1, B), dtype=torch.float32, device=self.device)
        right = torch.empty_like(left)
        _lib.check(_lib.lib().vbq_n_bit_intervals_f32(ops._ptr(z_cb), B, C, ops._ptr(self._table_dev()), N, ops._ptr(left),
                                                      ops._ptr(right), ops._stream(z_cb)), "vbq_n_bit_intervals_f32")
        return left, right

    # ------------------------------------------------------------------ the solve
    def _prep(self, batch_means, batch_spread, spread="sigma"):
        """Channel-major planes [C, B] of the means and the standard deviations, both in one launch; `spread` says what
        batch_spread holds ('sigma', or 'logvar': sigma = exp(.) ** 0.5 is taken on the way, quantizer.py:87,92)."""
        mu, sp = self._batch_dev(batch_means, batch_spread)
        if getattr(self, "validate_inputs", False):
            ops.check_inputs(mu.contiguous(), (sp if spread == "sigma" else torch.exp(sp) ** 0.5).contiguous())
        return ops.prep_planes(mu, sp, spread=spread)

    def _keyed_dev(self, name: str, arrays, builder):
        """Device copy of a table assembled from per-lambda model arrays, rebuilt only when one of those arrays is
        replaced.  Keyed on the arrays' identities; the cache entry holds the arrays, so an id cannot be reused by a
        new object while the entry lives.  An edit IN PLACE would leave the device copy stale without a trace, so the
        cached arrays are made read-only: such an edit now raises (NumPy's "assignment destination is read-only");
        replace the dict entry with a new array instead and the copy is rebuilt."""
        arrays = list(arrays)
        hit = self._dev_cache.get(name)
        if (hit is not None and len(hit[0]) == len(arrays) and all(a is b for a, b in zip(hit[0], arrays))
                and hit[1].device == self.device):
            return hit[1]
        t = builder().to(self.device)
        for a in arrays:
            if isinstance(a, np.ndarray):
                try:
                    a.setflags(write=False)
                except ValueError:
                    pass
        self._dev_cache[name] = (arrays, t)
        return t

    def _level_len_dev(self, lambs) -> Optional[torch.Tensor]:
        """quantizer.py:166,171-175: None for raw lengths, else f32 [L, C, N+1] = n + overhead."""
        rm = self.raw_code_length_entropy_models
        if not rm:
            return None
        if isinstance(rm, DeviceModels):       # still where the build wrote it: no host round trip
            t = rm.device_rows(lambs, "level_len")                                  # KeyError as in the reference
            if t is not None:
                return t
        arrays = [rm[lamb] for lamb in lambs]                                       # KeyError as in the reference
        return self._keyed_dev("level_len", arrays, lambda: self._level_len_host(lambs))

    def _level_len_host(self, lambs) -> torch.Tensor:
        N = self.max_bits_per_coord
        lv = np.arange(N + 1, dtype=np.int32).astype(np.float32)
        rows = []
        for lamb in lambs:
            model = np.asarray(self.raw_code_length_entropy_models[lamb])            # C x (N+1), KeyError as in the reference
            rows.append(lv[None, :].astype(model.dtype) + model)
        return torch.from_numpy(np.stack(rows).astype(np.float32))

    def _solve_idx(self, mu_cb, sg_cb, lambs, level_len):
        """u16 rank indices [L, C, B] on the device."""
        return ops.quantize(mu_cb, sg_cb, self._table_dev(), [float(l) for l in lambs], N=self.max_bits_per_coord,
                            level_len=level_len, layout="cb")

    def _workspace(self, name: str, nbytes: int) -> torch.Tensor:
        """A persistent device workspace, grown on demand (planes + index planes of the per-image call)."""
        ws = self._dev_cache.get(name)
        if ws is None or ws.numel() < nbytes or ws.device != self.device:
            ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
            self._dev_cache[name] = ws
        return ws

    def _stager(self):
        """The pinned staging blocks behind the lazy per-image results (vbq_amd.lazy.HostStager), one set per quantizer."""
        st = self._dev_cache.get("_stager")
        if st is None:
            from .lazy import HostStager
            st = self._dev_cache["_stager"] = HostStager()
        return st

    def _latents_call(self, means_bc, spread_bc, lambs, *, spread, level_len, models):
        """vbq_compress_latents_f32: planes, solve and the fused lookups of one batch in ONE C call (three launches).
        -> (Z_hat, raw_num_bits, num_bits | None), channel-last [L, B, C] device tensors."""
        from . import _lib
        N, C = self.max_bits_per_coord, self.num_channels
        if getattr(self, "validate_inputs", False):
            sig = spread_bc if spread == "sigma" else (torch.exp(spread_bc) ** 0.5 if spread == "logvar" else torch.sqrt(spread_bc))
            ops.check_inputs(means_bc.contiguous(), sig.contiguous())
        B = means_bc.shape[0]
        ws = self._workspace("_ws_latents", _lib.lib().vbq_compress_latents_workspace_bytes(B, C, len(lambs), N))
        return ops.compress_latents(means_bc, spread_bc, self._table_dev(), self._sorted_dev(), [float(l) for l in lambs], N=N,
                                    spread=spread, level_len=level_len, models=models, workspace=ws)

    def _batch_dev(self, batch_means, batch_stds):
        mu = torch.as_tensor(_to_numpy(batch_means) if not isinstance(batch_means, torch.Tensor) else batch_means)
        sg = torch.as_tensor(_to_numpy(batch_stds) if not isinstance(batch_stds, torch.Tensor) else batch_stds)
        mu = mu.to(self.device, torch.float32)
        sg = sg.to(self.device, torch.float32)
        if mu.dim() != 2 or mu.shape[1] != self.num_channels or mu.shape != sg.shape:
            raise ValueError(f"expected means/stds of shape [B, {self.num_channels}], got {tuple(mu.shape)} / {tuple(sg.shape)}")
        return mu, sg

    def compress_batch_channel_latents(self, batch_means, batch_stds, lambs, return_np=True, **kwargs):
        """quantizer.py:156-188 -> (Z_hat_dict, num_bits_dict), each dict[lamb] -> [B, C].
        num_bits is int32 (raw bit lengths) before the raw-length entropy models exist and
        float32 (n + overhead) afterwards, as in the reference."""
        lambs = list(lambs)
        mu, sg = self._batch_dev(batch_means, batch_stds)
        zhat, bits, _ = self._latents_call(mu, sg, lambs, spread="sigma", level_len=self._level_len_dev(lambs), models=None)
        if return_np:
            # ndarray-like views over the device tensors (vbq_amd.lazy): what is read on the host crosses PCIe, nothing else
            from .lazy import DeviceStack, group
            zs, bs = group([DeviceStack("batch_Z_hat", zhat, self._stager()), DeviceStack("batch_num_bits", bits, self._stager())])
            return dict(zip(lambs, zs.rows())), dict(zip(lambs, bs.rows()))
        return {lamb: zhat[i] for i, lamb in enumerate(lambs)}, {lamb: bits[i] for i, lamb in enumerate(lambs)}

    # ------------------------------------------------------------------ entropy models (quantizer.py:82-150)
    def _encode(self, X, vae):
        posterior_means, posterior_logvars = vae.encode(X)
        m = posterior_means if isinstance(posterior_means, torch.Tensor) else torch.from_numpy(_to_numpy(posterior_means))
        lv = posterior_logvars if isinstance(posterior_logvars, torch.Tensor) else torch.from_numpy(_to_numpy(posterior_logvars))
        return m.to(self.device, torch.float32), lv.to(self.device, torch.float32)

    def build_entropy_models(self, X, vae, lambs, add_n_smoothing):
        means, logvars = self._encode(X, vae)
        C = logvars.shape[-1]
        assert C == self.num_channels                                               # quantizer.py:89
        # batch_stds = exp(posterior_logvars) ** 0.5 (quantizer.py:87,92) is taken inside the planes kernel
        return self.build_entropy_models_from_latents(means.reshape(-1, C), logvars.reshape(-1, C), lambs, add_n_smoothing,
                                                      spread="logvar")

    def build_entropy_models_from_latents(self, batch_means, batch_stds, lambs, add_n_smoothing, spread="sigma"):
        """The body of quantizer.py:94-150 on (B x C) means/stds: vbq_amd.pipeline.EntropyModelBuild (pass 1 = solve +
        bit-length histogram in one kernel, length table on the device, pass 2 = solve + rank histogram + models).
        NOTHING here waits for the device: the three tables stay where the kernels wrote them behind dict-like views
        (`DeviceModels`) and reach NumPy when a caller reads them -- compress_latents never does.
        A rebuild starts from raw lengths again.  (In the reference a second build on the same object would feed the
        float "n + overhead" lengths of the first one to np.bincount, :96,104,166, which raises TypeError; starting
        over is the documented divergence)"""
        from .pipeline import EntropyModelBuild
        lambs = list(lambs)
        N, C = self.max_bits_per_coord, self.num_channels
        mu_bc, sp_bc = self._batch_dev(batch_means, batch_stds)
        if getattr(self, "validate_inputs", False):
            ops.check_inputs(mu_bc.contiguous(), (sp_bc if spread == "sigma" else torch.exp(sp_bc) ** 0.5).contiguous())
        B = mu_bc.shape[0]
        B_global, distributed = B, self.process_group is not None
        if distributed:
            import torch.distributed as dist
            t = torch.tensor([B], dtype=torch.int64, device=self.device)
            dist.all_reduce(t, group=self.process_group)
            B_global = int(t.item())
        # the index planes, the workspaces and the -log2 tables are kept between builds of one shape; the OUTPUT tensors are
        # new every time (the dicts of an earlier build keep theirs)
        build = EntropyModelBuild(B, C, [float(l) for l in lambs], self._table_dev(), N=N, add_n_smoothing=add_n_smoothing,
                                  global_rows=B_global, distributed=distributed, group=self.process_group,
                                  counts_dtype=torch.int32 if B_global < 2 ** 31 else torch.int64, keep_models=self._strict,
                                  buffers=self._dev_cache.setdefault("_build_buffers", {}))
        self.raw_code_length_entropy_models = None
        if build.one_call_ok and B > 0:
            # one GPU, tabulated -log2: planes, both passes, the histogram and the models behind ONE C call
            build.run_one_call(mu_bc, sp_bc, spread)
            level_len, raw_models, counts = build.level_len, build.raw_models, build.counts
        else:
            mu_cb, sg_cb = ops.prep_planes(mu_bc, sp_bc, spread=spread)
            build.pass1(mu_cb, sg_cb, None)
            level_len, raw_models = build.lengths()
            # pass 2: corrected lengths -> per-channel histogram of the code points (:118-148)
            _, counts = build.pass2(mu_cb, sg_cb, level_len)
        models_dev = build.finish_models()
        self._add_n_smoothing = add_n_smoothing
        self._dev_cache.pop("entropy_models", None)
        self._dev_cache.pop("level_len", None)
        if models_dev is not None:
            state = {"done": False}

            def deferred_checks():           # first host read of any of the three tables (synchronises)
                if not state["done"]:
                    state["done"] = True
                    build.check()

            if build.has_host_stages:        # a failed NumPy stage must not go unnoticed by device-only callers either
                deferred_checks()
            self.raw_code_length_entropy_models = DeviceModels(lambs, raw_models, {"level_len": level_len}, deferred_checks)
            self.entropy_models = DeviceModels(lambs, models_dev, None, deferred_checks)
            # kept for the entropy coder (vbq_amd.coder): the integer histograms behind the models
            self._code_counts = DeviceModels(lambs, counts, None, deferred_checks)
            return None
        # repeated f32 code points (or a model table too large to keep): the counts go to the host
        build.wait()
        torch.cuda.synchronize(self.device)
        build.check()
        raw_models = raw_models.cpu().numpy()
        self.raw_code_length_entropy_models = {lamb: raw_models[i] for i, lamb in enumerate(lambs)}
        counts = counts.cpu().numpy()
        if not self._strict:   # qidx is the FIRST sorted position of a repeated value (:135)
            merged = np.zeros_like(counts)
            for c in range(self.num_channels):
                np.add.at(merged[:, c, :], (slice(None), self._canon[c]), counts[:, c, :])
            counts = merged
        models = _entropy.neg_log2_freq(counts, add_n_smoothing)                    # [L, C, T] f32
        self.entropy_models = {lamb: models[i] for i, lamb in enumerate(lambs)}
        self._code_counts = {lamb: counts[i] for i, lamb in enumerate(lambs)}
        self._dev_cache["level_len"] = ([self.raw_code_length_entropy_models[lamb] for lamb in lambs], level_len)
        for d in (self.entropy_models, self.raw_code_length_entropy_models):
            for a in d.values():
                a.setflags(write=False)          # see _keyed_dev: in-place edits must not go unnoticed
        return None

    @property
    def lambs(self):
        return list(sorted(self.entropy_models.keys()))

    # ------------------------------------------------------------------ compression (quantizer.py:190-256)
    def _models_dev(self, lambs) -> torch.Tensor:
        """entropy_models of `lambs` as f32 [L, C, T] on the device, indexed by RANK."""
        em = self.entropy_models
        if isinstance(em, DeviceModels):
            t = em.device_rows(lambs)                                                # KeyError as in the reference
            if t is not None:
                return t

        def models_host():
            models = np.stack([np.asarray(em[lamb]) for lamb in lambs]).astype(np.float32)               # [L, C, T]
            if not self._strict:   # the reference indexes with the canonical qidx; same value either way once
                models = np.stack([np.take_along_axis(mm, self._canon, axis=1) for mm in models])        # mapped per rank
            return torch.from_numpy(models)
        return self._keyed_dev("entropy_models", [em[lamb] for lamb in lambs], models_host)

    def compress_latents(self, posterior_means, posterior_logvars, lambs, return_np=True):
        """quantizer.py:190-240 as ONE C call = three launches (vbq_compress_latents_f32: planes with the
        `exp(posterior_logvars) ** 0.5` of :197,202 folded in, the solve, and one pass over the indices that writes Z_hat,
        raw_num_bits and num_bits channel-last); no torch arithmetic anywhere in it.  return_np=False keeps the per-lambda results on the device (torch tensors shaped like the
        latents): no 150 MB device-to-host copy per Kodak image x 32 lambdas, which is what bounds the NumPy form (PCIe).
        The reference returns NumPy arrays (np.reshape moves to the CPU, :237); that is the default here too."""
        lambs = list(lambs)
        C = self.num_channels
        shape = tuple(np.shape(posterior_means))
        m = posterior_means if isinstance(posterior_means, torch.Tensor) else torch.from_numpy(_to_numpy(posterior_means))
        lv = posterior_logvars if isinstance(posterior_logvars, torch.Tensor) else torch.from_numpy(_to_numpy(posterior_logvars))
        m, lv = m.to(self.device, torch.float32), lv.to(self.device, torch.float32)
        assert lv.shape[-1] == C                                                     # quantizer.py:195
        # sigma = exp(logvar) ** 0.5 (quantizer.py:197,202) is taken inside the planes kernel
        zhat, raw_bits, num_bits = self._latents_call(m.reshape(-1, C), lv.reshape(-1, C), lambs, spread="logvar",
                                                      level_len=self._level_len_dev(lambs), models=self._models_dev(lambs))
        out_keys = ("Z_hat", "raw_num_bits", "num_bits_cl", "num_bits")
        output = {key: dict() for key in out_keys}
        L = len(lambs)
        if not return_np:
            arrs = {"Z_hat": zhat.reshape((L,) + shape), "raw_num_bits": raw_bits.reshape((L,) + shape),
                    "num_bits": num_bits.reshape((L,) + shape)}
        else:
            # NumPy-like views over the device tensors (vbq_amd.lazy): a quantity crosses PCIe -- through a persistent pinned
            # staging block, all its lambdas at once -- when somebody first READS it on the host; a caller that hands Z_hat back
            # to a decoder on the device and sums num_bits there (compress / vbq_amd.utils.evaluate_*) copies nothing.
            from .lazy import DeviceStack, group
            stager = self._stager()
            stacks = group([DeviceStack("Z_hat", zhat.reshape((L,) + shape), stager),
                            DeviceStack("raw_num_bits", raw_bits.reshape((L,) + shape), stager),
                            DeviceStack("num_bits", num_bits.reshape((L,) + shape), stager)])
            arrs = {st.name: st.rows() for st in stacks}
        has_cl = bool(self.raw_code_length_entropy_models)
        for i, lamb in enumerate(lambs):
            output["Z_hat"][lamb] = arrs["Z_hat"][i]
            output["raw_num_bits"][lamb] = arrs["raw_num_bits"][i]
            if has_cl:
                output["num_bits_cl"][lamb] = output["raw_num_bits"][lamb]          # :231-232
            output["num_bits"][lamb] = arrs["num_bits"][i]
        return output

    def compress_replay(self, X, vae, lambs, clip=True):
        """`compress(X, vae, lambs, clip)` AND what the evaluation loop reads of it (utils.py:547-556) as ONE HIP graph replay per
        image shape (vbq_amd.replay): -> (the result dict, (sums of num_bits, sums of num_bits_cl, uint8 X_hat) on the host).
        The dict's arrays live in the graph's static tensors: valid until the next call with the same shape.  Falls back to
        `compress` + `utils.evaluation_reads` when the VAE cannot be captured (NumPy VAEs, decoders that synchronise)."""
        from .replay import CompressReplay
        X = np.asarray(X) if not isinstance(X, np.ndarray) else X
        cache = self._dev_cache.setdefault("_replays", {})
        last = self._dev_cache.get("_replay_last")                # the loop's common case: the same shape, settings and VAE as last time
        if last is not None and last[0] == X.shape and last[1] is vae and last[2] == lambs and last[3] == bool(clip) and last[4] == X.dtype:
            key, rp = last[5], last[6]
        else:
            key = (X.shape, X.dtype.str, tuple(lambs), bool(clip), id(vae))
            rp = cache.get(key)
        fp = self._replay_fingerprint(lambs)
        if rp is None or rp.vae is not vae or rp.fingerprint != fp:
            if len(cache) >= 8:
                cache.clear()
            rp = CompressReplay(self, vae, X, lambs, clip)
            rp.fingerprint = self._replay_fingerprint(lambs)      # (after the capture: its warm-up may have grown the workspace)
            cache[key] = rp
        self._dev_cache["_replay_last"] = (X.shape, vae, list(lambs), bool(clip), X.dtype, key, rp)
        return rp.run(X)

    def _replay_fingerprint(self, lambs):
        """Addresses of everything a captured per-image graph reads besides its own static tensors: a new table, a rebuilt model
        or a workspace another shape made grow means a new capture."""
        ll, ws = self._level_len_dev(lambs), self._dev_cache.get("_ws_latents")
        return (self._table_dev().data_ptr(), self._sorted_dev().data_ptr(), None if ll is None else ll.data_ptr(),
                self._models_dev(lambs).data_ptr(), None if ws is None else (ws.data_ptr(), ws.numel()))

    # ------------------------------------------------------------------ real bits
    def codec(self, lambs, segment=1024):
        """rANS codec whose frequency tables are the histograms behind entropy_models[lamb]
        (one table per (lambda, channel)); see vbq_amd.coder."""
        from .coder import RansCodec, quantize_frequencies
        if self.entropy_models is None or not hasattr(self, "_code_counts"):
            raise ValueError("build_entropy_models() first")
        counts = np.stack([self._code_counts[lamb] for lamb in lambs])              # [L, C, T]
        freq = quantize_frequencies(counts, add_n_smoothing=self._add_n_smoothing)
        return RansCodec(freq.reshape(-1, self.quantization_levels), N=self.max_bits_per_coord, segment=segment)

    def encode_batch(self, batch_means, batch_stds, lambs, segment=1024):
        """Solve + entropy-code: returns (words, sizes, codec); total size = codec.compressed_bits(sizes).
        Streams are ordered [lambda][channel], each holding the B indices of that channel."""
        lambs = list(lambs)
        mu_cb, sg_cb = self._prep(batch_means, batch_stds)
        idx = self._solve_idx(mu_cb, sg_cb, lambs, self._level_len_dev(lambs))     # [L, C, B]
        if not self._strict:
            # repeated f32 code points: K1 may emit any rank of a run of equal values, the histogram behind the
            # frequency tables is the canonical one (first position, :135) -- code the canonical index (same Z_hat)
            canon = self._dev("canon", lambda: torch.from_numpy(self._canon))       # [C, T] int64
            idx = torch.gather(canon[None].expand(len(lambs), -1, -1), 2, idx.to(torch.int64)).to(torch.uint16)
        cdc = self.codec(lambs, segment)
        words, sizes = cdc.encode(idx)
        return words, sizes, cdc

    def decode_batch(self, words, sizes, codec, n_rows, lambs, return_np=True):
        """Inverse of encode_batch: dict[lamb] -> Z_hat [B, C]."""
        lambs = list(lambs)
        C = self.num_channels
        idx = codec.decode(words, sizes, n_rows).reshape(len(lambs), C, n_rows)
        zhat = ops.gather(idx, self._sorted_dev(), C, N=self.max_bits_per_coord, layout="cb", out_layout="bc")
        return {lamb: (zhat[i].cpu().numpy() if return_np else zhat[i]) for i, lamb in enumerate(lambs)}

    def compress(self, X, vae, lambs, clip=True):
        """quantizer.py:242-256.  With a torch VAE on the device nothing crosses PCIe here: the decoder gets the Z_hat tensor the
        kernels wrote (the reference -- and this method before round 5 -- went through NumPy: a device-to-host copy of L latent
        tensors and the same bytes back for `vae.decode`), and 'X_hat' comes back as lazy views like the other quantities."""
        from .lazy import DeviceStack, LazyArray, common_stack, join
        lambs = list(lambs)
        L = len(lambs)
        posterior_means, posterior_logvars = vae.encode(X)
        output = self.compress_latents(posterior_means, posterior_logvars, lambs)
        Z_hat_dict = output["Z_hat"]
        latent_shape = tuple(np.shape(posterior_means))
        z_dev = common_stack([Z_hat_dict[lamb] for lamb in lambs]) if isinstance(posterior_means, torch.Tensor) else None
        if z_dev is not None:
            Z_flat = z_dev.reshape((-1,) + latent_shape[1:])                          # len(lambs) * batch x latent shape[1:]
            if Z_flat.device != posterior_means.device:
                Z_flat = Z_flat.to(posterior_means.device)
        else:
            Z_hat_batch = np.stack([Z_hat_dict[lamb] for lamb in lambs])             # len(lambs) x latent shape
            Z_flat = Z_hat_batch.reshape((-1,) + latent_shape[1:])
            if isinstance(posterior_means, torch.Tensor):
                Z_flat = torch.from_numpy(Z_flat).to(posterior_means.device)
        X_hat = vae.decode(Z_flat)
        if isinstance(X_hat, torch.Tensor) and X_hat.is_cuda:
            X_hat_batch = X_hat.detach().reshape((L,) + tuple(np.shape(X)))
            if clip:
                X_hat_batch = X_hat_batch.clamp(0, 1)                                  # np.clip(X_hat_batch, 0, 1), :253
            xs = DeviceStack("X_hat", X_hat_batch.contiguous(), self._stager())
            if isinstance(Z_hat_dict[lambs[0]], LazyArray):
                join(Z_hat_dict[lambs[0]], xs)                                     # a quantity of THIS call (the prefetch rule)
            output["X_hat"] = dict(zip(lambs, xs.rows()))
            return output
        X_hat_batch = _to_numpy(X_hat).reshape((L,) + tuple(np.shape(X)))
        if clip:
            X_hat_batch = np.clip(X_hat_batch, 0, 1)
        output["X_hat"] = {lamb: X_hat_batch[i] for i, lamb in enumerate(lambs)}
        return output

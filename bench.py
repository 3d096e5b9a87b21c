#!/usr/bin/env python3
"""Headline benchmark of the VBQ hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--workload NAME]

One "step" = one whole entropy-model build of the reference (quantizer.py:82-150) over one batch, as
vbq_amd.pipeline.EntropyModelBuild runs it: [layout change] -> pass 1 (K1t: the 32-lambda solve with raw lengths as a
bit-length histogram, from 10 thresholds per element) -> length table -> pass 2 (K1: 32-lambda solve with corrected
lengths -> rank indices; K2: their per-(lambda, channel) histogram) -> code-length models, and, with N > 1, the RCCL
all-reduces of the two histograms.  `value` = (element, lambda) PAIRS quantized per second over all ranks -- one count per
pair and build, although a build solves every pair twice (the figure counting both passes is kept as
`solves_per_s_counting_both_passes`).  Inputs are resident in HBM before the timed region.

Workload at N = 1 (BASELINE.json configs[1]): the latent tensor of the Kodak-24 set from the paper's model
(--num_filters 256: 24 x 32 x 48 positions x 256 channels = 36864 x 256), 32-point lambda sweep
2**linspace(-8, 7.5, 32), per-channel code books of 2047 points.  Data are synthetic stand-ins of that shape (no
checkpoint / images ship with the reference).  N > 1: every rank owns its own batch of that shape (weak scaling; with
--scaling strong the rows of ONE tensor -- default synthetic_1e8, BASELINE configs[3] -- are split over the ranks).  All
ranks quantize against ONE code book: the per-channel second moments are all-reduced (ipynb:374 computed globally) and the
table follows from them; the only other collectives are the two histogram all-reduces.

Every timed region is exactly K steps between synchronisations (the graph-replay region is taken three times, the median
region is reported, all three are in the line); in front of it, after the W warm-up steps, the step is
repeated UNTIMED for 0.3 s (`clock_ramp`, VBQ_BENCH_RAMP_S) so that the timed steps run at the clock the device sustains under
this load -- after an idle gap the first ~50 ms of kernels run about 10 % slower, and W steps of 0.75 ms are over before that.

Prints ONE compact JSON line (rank 0, the LAST line of stdout, below 4 KB: `headline`) and writes the full record --
everything listed below in full -- to --full-record (default bench_full.json next to this file).  `python bench.py --gpus N`
started by hand starts its own N ranks (`launch_ranks`); under torch.distributed.run it is one of them.
  roofline       the dominant kernel (pass 2's K1, k_quant_fast): ALGORITHMIC bytes -- 8 B read per element + 2 B written
                 per (element, lambda) -- over its event-timed launches against 8 TB/s (bound = "hbm").  The committed
                 counters say what keeps it below that: VALU issue (`limited_by`, `valu_issue_frac`).
  rd_curve       rate (bits per latent from the entropy models, quantizer.py:226-228), distortion
                 sum((z - mu)^2 / (2 sigma^2)) / E and Lagrangian for every lambda of the sweep, on the whole tensor (f64
                 device reduction) and, on the parity sample, next to the oracle's (max_rel_diff <= 1e-5 asserted).
  cpu_baseline   the C oracle (oracle/vbq_oracle.c, OpenMP) on this host's cores on a bounded sample of the same workload.
  workloads      the other single-GPU configurations of BASELINE.json and the call patterns the reference itself uses
                 (the literal quantize(mu, sigma, lmbda) on channel-last latents, the 16-lambda build of
                 post_process.py:115, one lambda / one beta per call), each with its own roofline and oracle parity.
  per_gpu        (N > 1) every rank's kernel times and roofline fractions.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# the host driver only supports dmabuf IPC: RCCL between processes needs this before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, ROOT)

N_BITS = 10
T = 2 ** (N_BITS + 1) - 1
LAMBDAS = [float(v) for v in 2.0 ** np.linspace(-8, 7.5, 32)]
LAMBDAS_16 = [float(v) for v in 2.0 ** np.linspace(-8, 7, 16)]          # post_process.py:115
BETAS_50 = [float(b) for b in np.exp(np.linspace(np.log(0.01), np.log(100000), 50))]      # ipynb cell 32
HBM_PEAK = 8.0e12
HBM_COPY_MEASURED = 6.29e12            # float4 copy on MI355X (MI355X_MICROARCH.md)
XI = np.concatenate([(np.arange(2 ** n) + 0.5) / 2 ** n for n in range(N_BITS + 1)])

# DESIGN.md section 6, "What a SCALE run should show": the prediction table of the headline workload (weak scaling: one
# Kodak-24 tensor per rank, C = 256, 32 lambdas) under the key names the N > 1 line reports its OBSERVED values with
# (`scale_check`), so that the first real SCALE record can be diffed against it mechanically.  [lo, hi] ranges; the
# isolated all-reduce as [all links, one ring].
SCALE_PREDICTION = {
    2: {"rank_histogram_payload_bytes": 44.7e6, "allreduce_isolated_ms": [0.84, 0.84], "allreduce_hidden": False,
        "level_allreduce_exposed_ms": 0.045, "k1t_plus_k1_beside_collective_ms": [0.055, 0.060], "step_ms": [0.88, 0.95],
        "value": [6.4e11, 6.9e11], "efficiency_vs_n1": [0.68, 0.73]},
    4: {"rank_histogram_payload_bytes": 44.7e6, "allreduce_isolated_ms": [0.45, 1.26], "allreduce_hidden": True,
        "level_allreduce_exposed_ms": 0.055, "k1t_plus_k1_beside_collective_ms": [0.055, 0.060], "step_ms": [0.78, 0.84],
        "value": [1.44e12, 1.55e12], "efficiency_vs_n1": [0.76, 0.82]},
    8: {"rank_histogram_payload_bytes": 44.7e6, "allreduce_isolated_ms": [0.26, 1.48], "allreduce_hidden": True,
        "level_allreduce_exposed_ms": 0.070, "k1t_plus_k1_beside_collective_ms": [0.055, 0.060], "step_ms": [0.80, 0.86],
        "value": [2.8e12, 3.0e12], "efficiency_vs_n1": [0.74, 0.80]},
}


def scale_check(full):
    """`predicted` (headline workload only) next to `observed` (this run) under the same keys."""
    ar, per = full.get("allreduce"), full.get("per_gpu")
    if not ar or not per:
        return None
    n = full.get("n_gpus")
    beside = max((g.get("pass1_k1t_ms") or 0) + (g.get("pass2_k1_ms") or 0) - (g.get("pass1_k1t_ms_without_collectives") or 0)
                 - (g.get("pass2_k1_ms_without_collectives") or 0) for g in per)
    iso, step0 = ar.get("rank_histogram_allreduce_ms_isolated"), ar.get("ms_per_step_without_collectives")
    obs = {"rank_histogram_payload_bytes": ar.get("rank_histogram_payload_bytes"), "allreduce_isolated_ms": _sig(iso, 4),
           "allreduce_hidden": bool(iso is not None and step0 is not None and iso < step0),
           "level_allreduce_exposed_ms": _sig(ar.get("exposed_ms_per_step"), 3),
           "k1t_plus_k1_beside_collective_ms": _sig(beside, 3), "step_ms": _sig(full.get("ms_per_step"), 4),
           "value": _sig(full.get("value"), 4), "efficiency_vs_n1": None}      # the driver computes efficiency from its own N = 1 run
    headline_wl = str((full.get("config") or {}).get("workload", "")).startswith("kodak24_c256") and full.get("scaling") == "weak"
    return {"predicted": SCALE_PREDICTION.get(n) if headline_wl else None, "observed": obs}


WORKLOADS = {
    # name: (rows, channels, description)
    "kodak24_c256": (36864, 256, "Kodak-24 latents of bls2017 (num_filters=256): [36864 x 256] f32 (BASELINE.json configs[1])"),
    "kodak24_c32": (36864, 32, "Kodak-24 latents of bls2017 (default num_filters=32): [36864 x 32] f32"),
    "embeddings_1e7": (10_000_000, 1, "word embeddings 100000 x 100 (the notebook's own size, ipynb:169-172), one Gaussian code book"),
    "embeddings_4e5x300": (120_000_000, 1, "word embeddings 4e5 x 300 (BASELINE.json configs[2]), one Gaussian code book"),
    "synthetic_1e8": (100_000_000, 1, "synthetic 1e8-element tensor, one code book (BASELINE.json configs[3])"),
    "shard_1.25e8": (125_000_000, 1, "one rank's 1.25e8-element shard of the 1e9-element tensor (BASELINE.json configs[4])"),
}
# what the default run measures besides the headline workload (single-GPU configurations of BASELINE.json)
OTHER_WORKLOADS = ["kodak24_c32", "embeddings_1e7", "embeddings_4e5x300", "synthetic_1e8", "shard_1.25e8"]


def make_inputs(rows, C, seed, scale_seed=1000):
    """Synthetic (mu, sigma) with the statistics of SURVEY 8(d).  The per-channel spreads come from `scale_seed` (the same on
    every rank: the ranks hold different rows of one population), the values from `seed`."""
    s_c = np.exp(np.random.default_rng(scale_seed).uniform(np.log(0.3), np.log(3.0), C)) if C > 1 else np.array([1.2329])
    m_c = np.zeros(C) if C > 1 else np.array([-0.0799])
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((rows, C), dtype=np.float32)
    mu *= s_c.astype(np.float32)
    mu += m_c.astype(np.float32)
    sigma = rng.standard_normal((rows, C), dtype=np.float32)
    sigma *= np.float32(0.7)
    sigma -= np.float32(2.0)
    np.exp(sigma, out=sigma)
    np.clip(sigma, 1e-4, 10, out=sigma)
    return mu, sigma


def gaussian_tables(scale):
    """Level-major f32 [C, T]: norm.ppf(xi, scale = empirical std) per channel (ipynb:374,385; vae_models.py:40-43)."""
    from scipy.stats import norm
    return norm.ppf(XI[None, :], scale=np.asarray(scale, dtype=np.float64)[:, None]).astype(np.float32)


def make_inputs_with_table(rows, C, seed):
    """(mu, sigma, table) with the table's second moments summed on the host -- for the developer tools under tools/ (the
    bench itself takes the moments on the device and all-reduces them: empirical_tables)."""
    mu, sigma = make_inputs(rows, C, seed)
    scale = np.sqrt(np.array([np.add.reduce(mu[:, c].astype(np.float64) ** 2) / rows for c in range(C)]))
    return mu, sigma, gaussian_tables(scale)


def empirical_tables(x_dev, rows_local, C, layout):
    """ONE code book for all ranks: sqrt(mean x^2) per channel over every rank's rows (K3 moments on the device, the f64
    sums all-reduced -- ipynb:374 computed globally), then the Gaussian table on the host."""
    from vbq_amd import dist as vd, ops
    m = ops.moments(x_dev, layout=layout)                      # f64 [C, 2] = (sum x, sum x^2)
    return gaussian_tables(vd.global_empirical_std(m, rows_local))


def cpu_baseline(mu, sigma, tables, level_len, lambdas, target_s=12.0):
    from oracle import c_oracle as CO
    threads = CO.max_threads()
    rows, C = mu.shape
    L = len(lambdas)
    probe = max(1, min(rows, 200_000 // C))
    t0 = time.perf_counter()
    CO.quantize(mu[:probe], sigma[:probe], tables, lambdas, N=N_BITS, level_len=level_len, threads=threads)
    dt = time.perf_counter() - t0
    rate = probe * C * L / dt
    n = int(max(probe, min(rows, target_s * rate / (C * L))))
    reps = 0
    t0 = time.perf_counter()
    while reps == 0 or time.perf_counter() - t0 < target_s:      # the probe includes thread start-up: bound by time
        idx = CO.quantize(mu[:n], sigma[:n], tables, lambdas, N=N_BITS, level_len=level_len, threads=threads)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": reps * n * C * L / dt, "unit": "latents/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x (first {n} of {rows} rows x {C} channels x {L} lambdas, corrected lengths), C oracle "
                      f"(oracle/vbq_oracle.c, OpenMP {threads} threads), solve only, {dt:.1f} s"}, idx, n


def cpu_baseline_numpy(mu, sigma, tables, level_len, lambdas, idx_check, target_s=5.0, max_rows=2048):
    """The reference's own formulation in NumPy (per-level searchsorted on padded grids, 21 x B x C candidate
    tensors, per-lambda argmax: oracle/vbq_oracle.py restating quantizer.py:65-80,156-188 and utils.py:363-423;
    golden vectors G5/G6/G8 tie it to the reference's code), single process, NumPy's default threading."""
    from oracle import vbq_oracle as O
    rows, C = mu.shape
    L = len(lambdas)
    orc = O.ChannelwiseOracle(C, N_BITS)
    orc.build_code_points(lambda xi: tables.T)
    if level_len is not None:
        lv = np.arange(N_BITS + 1, dtype=np.float32)
        orc.raw_models = {lam: level_len[i] - lv[None, :] for i, lam in enumerate(lambdas)}
    n = max(1, min(rows, 64))
    t0 = time.perf_counter()
    orc.compress_batch(mu[:n], sigma[:n], lambdas)
    rate = n * C * L / (time.perf_counter() - t0)
    n = int(max(n, min(rows, max_rows, target_s * rate / (C * L))))      # the 21 x B x C x L length stack bounds B
    t0 = time.perf_counter()
    Z, _ = orc.compress_batch(mu[:n], sigma[:n], lambdas)
    dt = time.perf_counter() - t0
    ok = None
    m = min(n, 256)
    if idx_check is not None and idx_check.shape[1] >= m:
        q = np.stack([O.qidx_lookup(orc.by_channel, Z[i][:m]).T for i in range(L)])      # [L, m, C]
        ok = bool(np.array_equal(q.astype(np.uint16), idx_check[:, :m]))
    return {"value": n * C * L / dt, "unit": "latents/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"first {n} of {rows} rows x {C} channels x {L} lambdas, NumPy restatement of the reference formulation "
                      f"(oracle/vbq_oracle.py), one process, {dt:.1f} s", "agrees_with_c_oracle": ok}


def committed_counters(workload, rows, C, L):
    """HBM traffic / VALU counters of K1 per launch from the committed PMC passes (profiles/*_pmc.json: 2 x FETCH_SIZE +
    WRITE_SIZE, collected with rocprofv3 --pmc in separate runs of this same workload)."""
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
            pj = json.load(open(f))
            bl = pj.get("bench_line", {}).get("config", {})
            if bl.get("elements_per_gpu") == rows * C and bl.get("lambdas") == L and "k_quant_fast" in pj and \
                    bl.get("workload", "").startswith(workload) and "hbm_bytes_per_launch" in pj["k_quant_fast"]:
                src = os.path.relpath(f, ROOT)
                k = pj["k_quant_fast"]
                sqc = k.get("sq", {})
                valu = None
                if "SQ_INSTS_VALU" in sqc:          # why the HBM fraction is what it is: the kernel is VALU-issue-bound
                    valu = {"valu_wave_instructions_per_pass": sqc["SQ_INSTS_VALU"] * k.get("launches_per_pass", 1),
                            "valu_instructions_per_latent": sqc["SQ_INSTS_VALU"] * k.get("launches_per_pass", 1) * 64.0 / (rows * C * L),
                            "clock_GHz_under_load": sqc.get("clock_GHz"),
                            # share of the SIMDs' issue cycles spent on VALU instructions (quad-cycles per instruction
                            # weighted by the loop's mix of 2- and 4-cycle ops): the kernel's real ceiling
                            "valu_issue_frac": sqc.get("valu_issue_frac"), "source": src}
                return k["hbm_bytes_per_launch"] * k.get("launches_per_pass", 1), src, valu
    except Exception:
        pass
    return None, None, None


class Timers:
    """HIP events on whatever stream a pipeline stage runs on; per-step sums per stage."""

    def __init__(self, torch):
        self.torch = torch
        self.open = {}
        self.pairs = {}
        self.enabled = False

    def __call__(self, name, phase):
        if not self.enabled:
            return
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record(self.torch.cuda.current_stream())
        if phase == 0:
            self.open[name] = ev
        else:
            self.pairs.setdefault(name, []).append((self.open.pop(name), ev))

    def total_ms(self, name):
        return float(sum(a.elapsed_time(b) for a, b in self.pairs.get(name, [])))

    def launches(self, name):
        return len(self.pairs.get(name, []))


RAMP_S = float(os.environ.get("VBQ_BENCH_RAMP_S", "0.3"))


def clock_ramp(torch, fn, seconds=None, dist=None, dev=None):
    """UNTIMED: repeat `fn` for about `seconds` of wall time so that the timed region starts at the clock the device sustains
    under this load.  After an idle gap (process start, graph capture, host-side parity work) the first ~50 ms of kernels run
    measurably slower -- K1 0.397 ms against 0.355 ms once the clock has settled (same box, same build) -- and a handful of
    warm-up steps of 0.75 ms each is over before that.  With several ranks every rank runs the same number of repetitions (the
    step contains collectives): the count comes from the slowest rank's probe."""
    seconds = RAMP_S if seconds is None else seconds
    if seconds <= 0:
        return 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    per = max((time.perf_counter() - t0) / 3, 1e-6)
    if dist is not None:
        tt = torch.tensor([per], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        per = float(tt.item())
    n = int(min(5000, max(0, seconds / per)))
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return n + 3


def event_ms(torch, fn, steps, warmup):
    """Mean HIP-event time of `fn` (everything it enqueues on the current stream), after `warmup` calls and the clock ramp."""
    for _ in range(warmup):
        fn()
    clock_ramp(torch, fn, min(RAMP_S, 0.15))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))



def compact(o, keep=("value", "ms_per_step")):
    """The printed line: floats at 7 significant digits (the raw doubles doubled its length; `value`, `ms_per_step` and the
    oracle comparison's max_rel_diff keep theirs)."""
    if isinstance(o, dict):
        return {k: (v if k in keep and isinstance(v, float) else compact(v, keep)) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [compact(v, keep) for v in o]
    if isinstance(o, float) and o == o and abs(o) != float("inf"):
        return float(f"{o:.7g}")
    return o

LINE_LIMIT = 4096          # the driver keeps the tail of stdout: the LAST line must be one compact JSON object below this


def _sig(v, digits=6):
    if isinstance(v, float) and v == v and abs(v) != float("inf"):
        return float(f"{v:.{digits}g}")
    return v


def headline(full, side_file=None):
    """The ONE line printed on stdout: the contract's keys, `roofline` and `cpu_baseline` of the dominant kernel, the parity
    verdicts, and every other measured workload reduced to [ms_per_step, HBM fraction of its dominant kernel, parity].
    Everything else of `full` (R-D curve arrays, per-workload prose, per-rank reports) goes to the side file."""
    cfg = full.get("config") or {}
    roof = full.get("roofline") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": cfg.get("workload_short") or str(cfg.get("workload", ""))[:160],
                      "elements_per_gpu": cfg.get("elements_per_gpu"), "lambdas": cfg.get("lambdas"),
                      "parallelism": cfg.get("parallelism"), "launch": str(cfg.get("launch", ""))[:80],
                      "eager_ms_per_step_right_after_warmup": _sig(cfg.get("eager_ms_per_step_right_after_warmup")),
                      "untimed_clock_ramp_s": cfg.get("untimed_clock_ramp_s"),
                      # every region of exactly `steps` steps that was timed; `ms_per_step` is their median
                      "timed_regions_ms_per_step": [_sig(v, 4) for v in (full.get("timed_regions_ms_per_step") or [])]}
    line["roofline"] = {k: _sig(roof.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                        "algorithmic_bytes_per_launch", "avg_launch_ms", "limited_by",
                                                        "valu_issue_frac", "traffic_source", "frac_of_measured_copy", "ceiling_frac",
                                                        "step_intermediate_bytes")}
    line["roofline"]["kernel"] = str(roof.get("kernel", "")).split(" ")[0]
    if roof.get("traffic_source"):                           # "<file> (committed ..., not this run)" -> file + tag
        line["roofline"]["traffic_source"] = str(roof["traffic_source"]).split(" ")[0] + " (committed pmc, not this run)"
    line["roofline"]["per_lambda_read_frac"] = [_sig(v, 3) for v in (roof.get("per_lambda_read_frac") or [])] or None
    cb = full.get("cpu_baseline")
    line["cpu_baseline"] = None if not cb else {"value": _sig(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                                "kind": cb.get("kind"), "sample": str(cb.get("sample", ""))[:200]}
    line["parity_vs_oracle_on_sample"] = full.get("parity_vs_oracle_on_sample")
    line["rd_lagrangian_max_rel_diff_vs_oracle"] = _sig(((full.get("rd_curve") or {}).get("vs_oracle_on_sample") or {}).get("max_rel_diff"), 3)
    st = full.get("stages_ms") or {}
    line["stages_ms"] = {short: _sig(v, 4) for short, v in zip(("layout", "k1t", "k1", "k2"), st.values())} if st else None
    # HBM fraction of every stage's kernel on ITS algorithmic bytes (K1t: 8 B / element read; K1: 8 B / element + 2 B / pair;
    # K2: 2 B / pair read), event-timed in this run: all three recomputable from the line alone
    line["stages_frac"] = {"k1t": _sig((full.get("roofline_k1h") or {}).get("frac"), 3), "k1": _sig(roof.get("frac"), 3),
                           "k2": _sig((full.get("roofline_k2_histogram") or {}).get("frac"), 3)}
    if full.get("allreduce"):
        ar = full["allreduce"]
        line["allreduce"] = {"payload_bytes": ar.get("rank_histogram_payload_bytes"), "packed_3x21": ar.get("packed_3x21"),
                             "isolated_ms": _sig(ar.get("rank_histogram_allreduce_ms_isolated"), 4),
                             "ms_per_step_without_collectives": _sig(ar.get("ms_per_step_without_collectives"), 4),
                             "exposed_ms_per_step": _sig(ar.get("exposed_ms_per_step"), 4)}
    sc = scale_check(full)
    if sc:                           # DESIGN section 6's prediction and this run, same keys
        line["scale_check"] = sc
    if full.get("per_gpu"):          # [k1t ms, k1 ms, k2 ms, K1 HBM fraction, k1 ms in the steps without collectives] per rank
        line["per_gpu"] = [[_sig(g.get("pass1_k1t_ms"), 4), _sig(g.get("pass2_k1_ms"), 4), _sig(g.get("pass2_k2_ms"), 4),
                            _sig(g.get("k1_hbm_frac"), 3), _sig(g.get("pass2_k1_ms_without_collectives"), 4)] for g in full["per_gpu"]]
    if full.get("workloads"):                                # name: [ms_per_step, HBM fraction of its dominant kernel, parity]
        line["workloads"] = {k: [_sig(v.get("ms_per_step"), 4), _sig((v.get("roofline") or {}).get("frac"), 3),
                                 v.get("parity_ok", v.get("parity_vs_oracle_on_sample"))] for k, v in full["workloads"].items()}
    line["full_record"] = side_file
    # never let the line outgrow the driver's parser again: shed the optional parts, largest first
    for drop in ("workloads", "per_gpu", "allreduce", "scale_check", "stages_ms", "stages_frac"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(drop, None)
    return line


def emit(full, path):
    """Write the full record to `path` (best effort) and print the compact line as the LAST line of stdout."""
    full = compact(full)
    side = None
    if path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as f:
                json.dump(full, f)
            side = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
        except OSError as e:
            print(f"bench.py: could not write the full record to {path}: {e}", file=sys.stderr)
    sys.stdout.flush()
    print(json.dumps(headline(full, side)), flush=True)


def launch_ranks(n):
    """`python bench.py --gpus N` started by hand (no WORLD_SIZE in the environment): this process -- which has made no GPU
    call and has not imported torch -- starts the N ranks as child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    relays rank 0's line as its own last line and exits non-zero if any rank did.  Never re-execs."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.extend(procs[0].stdout), daemon=True)      # drain rank 0's stdout as it comes
    reader.start()
    # A rank that dies early (no such device, a failed build) would leave the others waiting in the rendezvous for many
    # minutes: watch all of them and end the rest -- these exact children, nothing else -- as soon as one has failed.
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            deadline = time.time() + 5.0                       # let the others fail on their own first (their messages matter)
            while time.time() < deadline and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out0 = "".join(buf)
    if failed:
        out0 = "\n".join(l for l in out0.splitlines() if not l.startswith("{"))      # no record from a failed job
    lines = [l for l in (out0 or "").splitlines() if l.strip()]
    for l in (lines if failed else lines[:-1]):
        print(l, file=sys.stderr)
    if lines and not failed:
        print(lines[-1], flush=True)
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(f"bench.py --gpus {n}: ranks failed (rank, exit code): {bad}")


def rd_curve(ops, mu, sg, idx, tab_h, models, C, layout, lambdas, E):
    """Rate / distortion / Lagrangian per lambda of the indices `idx` on the device (vbq_rd_sums_u16, f64)."""
    import torch
    srt = torch.from_numpy(np.sort(tab_h, axis=1)).to(mu.device)
    s = ops.rd_sums(mu, sg, idx, srt, C, N=N_BITS, layout=layout, rate=models).cpu().numpy()
    dist_, rate = s[:, 0] / E, s[:, 1] / E
    return {"distortion": dist_, "rate": rate, "lagrangian": dist_ + np.asarray(lambdas) * rate}


def rd_curve_oracle(mu_w, sg_w, idx_w, tab_h, models_h, lambdas):
    """The same three numbers from the oracle's indices on a window of rows: idx_w [L, rows, C], NumPy f64."""
    srt = np.sort(tab_h, axis=1).astype(np.float64)
    C = mu_w.shape[1]
    ch = np.arange(C)[None, :]
    n = mu_w.size
    d, r = [], []
    for l in range(len(lambdas)):
        q = idx_w[l].astype(np.int64)
        z = srt[ch, q]
        d.append(float(np.sum((z - mu_w.astype(np.float64)) ** 2 / (2.0 * sg_w.astype(np.float64) ** 2))) / n)
        r.append(float(np.sum(models_h[l][ch, q].astype(np.float64))) / n)
    d, r = np.array(d), np.array(r)
    return {"distortion": d, "rate": r, "lagrangian": d + np.asarray(lambdas) * r}


def run_workload(name, args, torch, dist, dev, rank, world, steps, warmup, detailed, lambdas=None, strong=False):
    """Time `steps` alternations of one workload; returns the result dict (rank 0) or None."""
    from vbq_amd import dist as vd, entropy, ops
    from vbq_amd.pipeline import EntropyModelBuild
    lambdas = LAMBDAS if lambdas is None else lambdas
    rows_total, C, desc = WORKLOADS[name]
    if strong:                                                   # rows of ONE tensor split contiguously over the ranks
        r0, r1 = vd.shard_rows(rows_total, rank, world)
        rows, global_rows = r1 - r0, rows_total
    else:                                                        # every rank owns a batch of the workload's shape
        rows, global_rows = rows_total, rows_total * world
    L = len(lambdas)
    E = rows * C
    mu_h, sg_h = make_inputs(rows, C, seed=1000 + rank)
    # Channel-last [rows, C] is how the latents arrive (quantizer.py:90-91).  The kernels work on channel-major
    # planes [C, rows]; the layout change is part of every timed step.
    if C > 1:
        mu_in, sg_in = torch.from_numpy(mu_h).to(dev), torch.from_numpy(sg_h).to(dev)
        mu = torch.empty((C, rows), dtype=torch.float32, device=dev)
        sg = torch.empty((C, rows), dtype=torch.float32, device=dev)
        tab_h = empirical_tables(mu_in, rows, C, "bc")
    else:
        mu_in = sg_in = None
        mu, sg = torch.from_numpy(mu_h.reshape(1, rows)).to(dev), torch.from_numpy(sg_h.reshape(1, rows)).to(dev)
        tab_h = empirical_tables(mu.reshape(rows), rows, 1, "bc")
    tab = torch.from_numpy(tab_h).to(dev)
    build = EntropyModelBuild(rows, C, lambdas, tab, N=N_BITS, add_n_smoothing=1, global_rows=global_rows,
                              distributed=world > 1, level_group=args.level_group, n_chunks=args.chunks)
    if os.environ.get("VBQ_K1_WG_PER_CU"):
        build.k1_workgroups_per_cu = int(os.environ["VBQ_K1_WG_PER_CU"])
    timers = Timers(torch)
    build.timers = timers

    def step():
        if mu_in is not None:
            timers("layout", 0)
            ops.prep_planes(mu_in, sg_in, out_mu=mu, out_sigma=sg)      # both layout changes in one launch
            timers("layout", 1)
        build.run(mu, sg)

    def timed(n):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        build.wait()
        if world > 1:
            build.finish_models()          # the last step's model table (sharded builds look it up one step late)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    for _ in range(warmup):
        step()
    build.wait()
    cold_ms = timed(steps) / steps * 1e3     # the contract's letter -- W warm-up steps, then K steps -- kept for comparison
    ramp_steps = clock_ramp(torch, step, dist=dist if world > 1 else None, dev=dev)      # untimed, see clock_ramp
    build.wait()
    torch.cuda.synchronize()
    timers.enabled = True
    dt = timed(steps)
    timers.enabled = False
    build.check()                      # the LUT assumption and the packed counters' overflow guard, outside the timed loop
    eager_ms = dt / steps * 1e3
    timed_regions_ms = [eager_ms]
    launch = "eager launches, two streams" if len(build.chunks) > 1 else "eager launches"
    # The step only enqueues stream-ordered work (no allocation, no host synchronisation: the -log2 step is a table
    # lookup or a stream-ordered host callback), so the whole alternation replays from one captured HIP graph: ~20
    # launches become one.  Per-kernel times above come from the eager steps (events cannot sit inside a graph); the
    # step time from the replays.
    if args.graph and world == 1 and build.graph_safe:
        try:
            graph = torch.cuda.CUDAGraph()
            cs = torch.cuda.Stream()
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                step()
            torch.cuda.current_stream().wait_stream(cs)
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                step()
            for _ in range(max(3, warmup)):
                graph.replay()
            clock_ramp(torch, graph.replay)            # untimed: capture left the device idle
            # The timed region -- EXACTLY `steps` replays between two synchronisations -- is taken three times back to back and
            # the MEDIAN region is the one reported (all three are in the full record): one region of 13 ms is short enough
            # for a transient on the box to land in it (0.81 instead of 0.63 ms per step was seen once in this round).
            regions = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    graph.replay()
                torch.cuda.synchronize()
                regions.append(time.perf_counter() - t0)
            dt = sorted(regions)[1]
            timed_regions_ms = [r / steps * 1e3 for r in regions]
            launch = f"one HIP graph replay per step (eager: {eager_ms:.3f} ms)"
        except Exception as e:                     # the eager measurement above stands
            torch.cuda.synchronize()
            launch += f" (HIP graph capture failed: {type(e).__name__})"
    k1_ms, k1_n = timers.total_ms("k1") / steps, timers.launches("k1") // steps
    k1h_ms = timers.total_ms("k1h") / steps
    k2_ms = timers.total_ms("k2") / steps
    alg_bytes = E * (8 + 2 * L)                              # one K1 pass: 8 B in per element, 2 B out per solve
    traffic, traffic_src, valu = committed_counters(name, rows, C, L)
    total_E = E * world if not strong else rows_total * C
    res = {
        "ms_per_step": dt / steps * 1e3,
        # every timed region of exactly `steps` steps that was taken (graph replays: three, the median is `ms_per_step`)
        "timed_regions_ms_per_step": timed_regions_ms,
        # one count per (element, lambda) pair and build
        "value": total_E * L * steps / dt,
        "pairs_per_step": total_E * L,
        # the same time counting both solves of every pair (pass 1 solves for the bit-length histogram only)
        "solves_per_s_counting_both_passes": 2 * total_E * L * steps / dt,
        "roofline": {"bound": "hbm", "limited_by": "valu issue", "kernel": "k_quant_fast (pass 2: corrected lengths -> rank indices)",
                     "achieved": alg_bytes / (k1_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": alg_bytes / (k1_ms * 1e-3) / HBM_PEAK, "traffic": traffic,
                     # `traffic` and `valu_issue_frac` are NOT measured in this run: they are the newest committed rocprofv3 --pmc
                     # passes of this same workload (2 x FETCH_SIZE + WRITE_SIZE per launch), named here
                     "traffic_source": (traffic_src + " (committed rocprofv3 --pmc passes, not this run)") if traffic_src else None,
                     # the same achieved rate against the 6.29 TB/s a float4 copy measures on this chip
                     "frac_of_measured_copy": alg_bytes / (k1_ms * 1e-3) / HBM_COPY_MEASURED,
                     # what the exact dense argmin can reach with a perfect locate: VALU issue slots, not bytes
                     "ceiling_frac": 0.35, "ceiling_source": "EXPERIMENTS.md, 'a bound for the dense formulation' (rounds 1-2 text) and 'K1's phase B, costed' (round 5)",
                     # throughput against the north star's unfused per-lambda ceiling (8 B read per latent): [this kernel, whole step]
                     "per_lambda_read_frac": [E * L / (k1_ms * 1e-3) / (HBM_PEAK / 8.0), E * L * steps / dt / (HBM_PEAK / 8.0)],
                     # the index planes K1 writes and K2 reads back: intermediate bytes of the STEP that are in no result
                     "step_intermediate_bytes": 4 * L * E,
                     "step_floor_bytes": 16 * E + 8 * L * C * T,
                     "algorithmic_bytes_per_pass": alg_bytes, "launches_per_pass": k1_n,
                     "algorithmic_bytes_per_launch": alg_bytes / max(k1_n, 1), "avg_launch_ms": k1_ms / max(k1_n, 1),
                     "pass_ms": k1_ms, "latents_per_s_kernel_only": E * L / (k1_ms * 1e-3),
                     # what bounds the kernel: share of the SIMDs' issue cycles spent on its VALU instructions (committed
                     # SQ counters); achieved / peak / frac are the HBM figures that issue rate moves
                     "valu_issue_frac": (valu or {}).get("valu_issue_frac"), "valu": valu,
                     "bound_note": "HBM roofline on algorithmic bytes; what limits the kernel below it is instruction issue (VALU) "
                                   "-- counters in `valu`",
                     # the north star's "HBM-read roofline" prices the UNFUSED per-lambda call, 8 B read per latent,
                     # i.e. a ceiling of 1.0e12 latents/s at 8 TB/s (target 60 % = 6.0e11).  The fused kernels read every element
                     # once per sweep, so this is a throughput ratio against that ceiling, not a bandwidth fraction.
                     "per_lambda_read_roofline": {"bytes_read_per_latent": 8, "ceiling_latents_per_s": HBM_PEAK / 8.0,
                                                  "kernel_frac": E * L / (k1_ms * 1e-3) / (HBM_PEAK / 8.0),
                                                  "step_frac_per_gpu": E * L * steps / dt / (HBM_PEAK / 8.0)},
                     "note": "K2 of the previous row chunk runs concurrently on a second stream" if k1_n > 1 else None},
        "stages_ms": {"layout_change": timers.total_ms("layout") / steps if C > 1 else None,
                      "pass1_k1t_solve_and_level_histogram": k1h_ms, "pass2_k1_solve": k1_ms,
                      "pass2_k2_histogram (overlapped with k1)" if k1_n > 1 else "pass2_k2_histogram": k2_ms},
        "roofline_k1h": {"bound": "hbm", "limited_by": "valu issue", "kernel": "k_level_counts_hull (pass 1, K1t: thresholds instead of a lambda loop; no per-element output)",
                         "achieved": 8.0 * E / (k1h_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": 8.0 * E / (k1h_ms * 1e-3) / HBM_PEAK, "algorithmic_bytes_per_launch": 8 * E,
                         "avg_launch_ms": k1h_ms, "latents_per_s_kernel_only": E * L / (k1h_ms * 1e-3),
                         "note": "reads 8 B per element and writes nothing per element: instruction- and LDS-bound by construction"},
        "roofline_k2_histogram": {"bound": "hbm", "kernel": "k_hist_flat", "achieved": 2.0 * L * E / (k2_ms * 1e-3) / 1e9,
                                  "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": 2.0 * L * E / (k2_ms * 1e-3) / HBM_PEAK,
                                  "algorithmic_bytes_per_pass": 2 * L * E, "pass_ms": k2_ms},
        "config": {"workload": f"{name}: {desc}; {L}-point lambda sweep {sweep_name(lambdas)}; N={N_BITS} (2047 code points/"
                               f"channel); step = two-pass entropy-model build (quantizer.py:82-150): "
                               f"{'layout change + ' if C > 1 else ''}pass 1 (solve, raw lengths, bit-length histogram) + length table + "
                               f"pass 2 (solve, corrected lengths, rank indices + rank histogram"
                               f"{', K2 overlapped in ' + str(len(build.chunks)) + ' row chunks' if len(build.chunks) > 1 else ''}) + models"
                               f"{' + RCCL all-reduce of both histograms (the model table of a step is looked up while the next step runs)' if world > 1 else ''}; one code book for all ranks "
                               f"(second moments all-reduced)" + ("; rows of one tensor split over the ranks" if strong else ""),
                   "workload_short": f"{name} [{rows} x {C}] f32, {L}-lambda sweep, N={N_BITS}; step = two-pass entropy-model build "
                                     f"(quantizer.py:82-150)" + ("; rows of one tensor split over the ranks" if strong else ""),
                   "elements_per_gpu": E, "lambdas": L, "parallelism": f"element-sharded x{world}",
                   "launch": launch, "untimed_clock_ramp_s": RAMP_S,
                   "clock_ramp": f"{RAMP_S} s of untimed steps ({ramp_steps} eager steps here) in front of every timed region, after the "
                                 f"{warmup} warm-up steps: the timed {steps} steps run at the clock the device sustains under this load",
                   "eager_ms_per_step_right_after_warmup": cold_ms,
                   "length_table": build.length_table_route, "models": build.models_route},
    }
    if world > 1:
        # what the collectives cost: the same steps without them, and the rank-histogram all-reduce in isolation
        build.collectives = False
        for _ in range(2):
            step()
        timers0 = Timers(torch)
        timers0.enabled = True
        build.timers = timers0
        n0 = max(3, steps // 3)
        dt0 = timed(n0) / n0
        build.timers = timers
        build.collectives = True
        torch.cuda.synchronize()
        reps = 5
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if build.reducer is not None:
                build.reducer.start(build.counts).wait()
            else:
                dist.all_reduce(build.counts)
        torch.cuda.synchronize()
        ar = (time.perf_counter() - t0) / reps
        payload = build.reducer.payload_bytes(build.counts) if build.reducer is not None else build.counts.numel() * build.counts.element_size()
        res["allreduce"] = {"rank_histogram_payload_bytes": payload, "level_histogram_payload_bytes": build.level_counts.numel() * 8,
                            "packed_3x21": bool(build.reducer is not None and build.reducer.packed),
                            "counter_dtype": str(build.counts.dtype).replace("torch.", ""),
                            "rank_histogram_allreduce_ms_isolated": ar * 1e3,
                            "ms_per_step_without_collectives": dt0 * 1e3,
                            "exposed_ms_per_step": max(0.0, dt / steps - dt0) * 1e3,
                            "reserved_workgroups": build.reserved_workgroups,
                            "overlap": "asynchronous, two buffers: step i's rank histogram is reduced while step i+1 computes; "
                                       "the bit-length histogram has its own communicator"}
        # one last step with the collectives on, so that the buffers checked below hold GLOBAL histograms
        step()
        build.wait()
        torch.cuda.synchronize()
        # every rank's own timers (configs[4]: "per-GPU roofline report")
        mine = {"rank": rank, "elements": E, "pass1_k1t_ms": k1h_ms, "pass2_k1_ms": k1_ms, "pass2_k2_ms": k2_ms,
                # the same kernels in the steps run WITHOUT the collectives (every slot of the chip theirs): what the all-reduce
                # beside them costs the kernels themselves, per rank
                "pass1_k1t_ms_without_collectives": timers0.total_ms("k1h") / n0,
                "pass2_k1_ms_without_collectives": timers0.total_ms("k1") / n0,
                "pass2_k2_ms_without_collectives": timers0.total_ms("k2") / n0,
                "reserved_workgroups": build.reserved_workgroups,
                "layout_change_ms": timers.total_ms("layout") / steps if C > 1 else None,
                "k1_hbm_frac": alg_bytes / (k1_ms * 1e-3) / HBM_PEAK, "k1t_hbm_frac": 8.0 * E / (k1h_ms * 1e-3) / HBM_PEAK,
                "k2_hbm_frac": 2.0 * L * E / (k2_ms * 1e-3) / HBM_PEAK,
                "pairs_per_s_kernels_only": E * L / ((k1h_ms + k1_ms + k2_ms) * 1e-3)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        res["per_gpu"] = gathered

    if not detailed:
        if rank == 0:                     # what the CPU baseline of the line needs (no index planes: nothing to cross-check against)
            torch.cuda.synchronize()
            res["_host"] = (mu_h, sg_h, tab_h, build.level_len.cpu().numpy(), None)
        del build
        torch.cuda.empty_cache()
        return res if rank == 0 else None

    # ------------------------------------------------------------ parity of the timed configuration (every rank)
    from oracle import c_oracle as CO, vbq_oracle as O
    th = max(1, CO.max_threads() // world)
    torch.cuda.synchronize()
    ll_h = build.level_len.cpu().numpy()
    parity = {}
    # (1) pass-2 indices against the C oracle on windows at the start, across the middle and at the end of THIS rank's rows
    win = min(rows, max(1, 200_000 // C))
    starts = sorted({0, max(0, rows // 2 - win // 2), rows - win})
    ok = True
    for s in starts:
        want = CO.quantize(mu_h[s:s + win], sg_h[s:s + win], tab_h, lambdas, N=N_BITS, level_len=ll_h, threads=th)   # [L, win, C]
        got = build.idx[:, :, s:s + win].permute(0, 2, 1).cpu().numpy()
        ok = ok and bool(np.array_equal(got, want))
    parity["pass2_indices_equal_oracle_on_windows"] = ok
    parity["windows"] = f"{len(starts)} x {win} rows x {C} channels x {L} lambdas" + (" on every rank" if world > 1 else "")
    # (2) both histograms against a second route on the device, full size: K1 indices (raw lengths) -> K2 -> level sums
    #     must equal pass 1's K1t counts; K2 of the stored pass-2 indices in one launch must equal the build's counts.
    #     With N > 1 the build's buffers hold the all-reduced histograms: the second route is summed over the ranks too.
    lc = build.level_counts.clone()
    cnt = build.counts.clone().to(torch.int64)
    idx_raw = ops.quantize(mu, sg, tab, lambdas, N=N_BITS, layout="cb")
    via = entropy.level_counts_from_counts(ops.histogram(idx_raw, C, N=N_BITS, layout="cb"), N_BITS)
    del idx_raw
    one = ops.histogram(build.idx, C, N=N_BITS, layout="cb")
    if world > 1:
        dist.all_reduce(via)
        dist.all_reduce(one)
    parity["pass1_level_counts_equal_k1_k2_route" + ("_summed_over_ranks" if world > 1 else "")] = bool(torch.equal(via, lc))
    parity["pass2_rank_counts_equal_single_launch" + ("_summed_over_ranks" if world > 1 else "")] = bool(torch.equal(one, cnt))
    parity["counts_total"] = bool(int(cnt.sum().item()) == global_rows * C * L and int(lc.sum().item()) == global_rows * C * L)
    # (3) the length table and the models against the reference's NumPy float32 arithmetic on the same (global) counts
    parity["length_table_equals_numpy"] = bool(np.array_equal(
        ll_h, (np.arange(N_BITS + 1, dtype=np.float32) + entropy.neg_log2_freq(lc, 1)).astype(np.float32)))
    models_np = entropy.neg_log2_freq(cnt, 1)
    models_dev = build.finish_models()
    if models_dev is not None:
        parity["models_equal_numpy"] = bool(np.array_equal(models_dev.cpu().numpy(), models_np))
    # (4) level counts of pass 1 against the oracle on a sample (exact on the first window's rows)
    w0 = O.levels_of_sorted_ranks(N_BITS)[CO.quantize(mu_h[:win], sg_h[:win], tab_h, lambdas, N=N_BITS, threads=th)]
    want_lc = np.stack([[np.bincount(w0[l, :, c], minlength=N_BITS + 1) for c in range(C)] for l in range(L)])
    got_lc = ops.level_counts(mu[:, :win].contiguous(), sg[:, :win].contiguous(), tab, lambdas, N=N_BITS, layout="cb")
    parity["pass1_level_counts_equal_oracle_on_sample"] = bool(np.array_equal(got_lc.cpu().numpy(), want_lc))
    # (5) the R-D curve (BASELINE metric): whole tensor on the device; on the first window next to the oracle's
    models_t = models_dev if models_dev is not None else torch.from_numpy(models_np).to(dev)
    full = rd_curve(ops, mu, sg, build.idx, tab_h, models_t, C, "cb", lambdas, E)
    gpu_w = rd_curve(ops, mu[:, :win].contiguous(), sg[:, :win].contiguous(), build.idx[:, :, :win].contiguous(), tab_h,
                     models_t, C, "cb", lambdas, win * C)
    want0 = CO.quantize(mu_h[:win], sg_h[:win], tab_h, lambdas, N=N_BITS, level_len=ll_h, threads=th)
    orc_w = rd_curve_oracle(mu_h[:win], sg_h[:win], want0, tab_h, models_np, lambdas)
    rel = max(float(np.max(np.abs(gpu_w[k] - orc_w[k]) / np.maximum(np.abs(orc_w[k]), 1e-300))) for k in ("rate", "distortion", "lagrangian"))
    # (asserted at the end of main(), after every rank has passed the collectives below: a rank that raised here would leave the
    #  others waiting in all_gather_object)
    res["rd_curve"] = {"lambda": [float(v) for v in lambdas],
                       "rate_bits_per_latent": [float(v) for v in full["rate"]],
                       "distortion_per_latent": [float(v) for v in full["distortion"]],
                       "lagrangian_per_latent": [float(v) for v in full["lagrangian"]],
                       "scope": f"all {E} elements of this rank, pass-2 indices, f64 device reduction (vbq_rd_sums_u16); rate = "
                                f"entropy-model bits (quantizer.py:226-228), distortion = (z - mu)^2 / (2 sigma^2)",
                       "vs_oracle_on_sample": {"rows": win, "gpu_lagrangian": [float(v) for v in gpu_w["lagrangian"]],
                                               "oracle_lagrangian": [float(v) for v in orc_w["lagrangian"]],
                                               "gpu_rate": [float(v) for v in gpu_w["rate"]], "oracle_rate": [float(v) for v in orc_w["rate"]],
                                               "max_rel_diff": rel, "tolerance": 1e-5}}
    parity["rd_curve_within_1e-5_of_oracle"] = bool(rel <= 1e-5)
    ok_all = all(v for v in parity.values() if isinstance(v, bool))
    if world > 1:                                                # every rank's verdict
        flags = [None] * world
        dist.all_gather_object(flags, {"rank": rank, "ok": ok_all, "failed": [k for k, v in parity.items() if v is False]})
        parity["ranks"] = flags
        ok_all = all(f["ok"] for f in flags)
    res["parity"] = parity
    res["parity_ok"] = ok_all
    res["_host"] = (mu_h, sg_h, tab_h, ll_h, build.idx)
    return res if rank == 0 else None


def sweep_name(lambdas):
    if lambdas is LAMBDAS or list(lambdas) == LAMBDAS:
        return "2**linspace(-8,7.5,32)"
    if list(lambdas) == LAMBDAS_16:
        return "2**linspace(-8,7,16) (post_process.py:115)"
    return f"{len(lambdas)} values"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank owns a batch of the workload's shape (default kodak24_c256); strong: the rows of one "
                         "tensor (default synthetic_1e8, BASELINE configs[3]) are split over the ranks")
    ap.add_argument("--chunks", type=int, default=None, help="row chunks of pass 2 (K2 of chunk j overlaps K1 of chunk j+1 on a second "
                                                             "stream); default 1: the overlap measured slower, profiles/r2_overlap_sweep.txt")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="time eager launches instead of replaying the step from a captured HIP graph (one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the `workloads` section (the other BASELINE configs)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity block of the N > 1 line (the N = 1 line always has it)")
    ap.add_argument("--notebook", action="store_true",
                    help="C = 1 workloads in the word-embedding notebook's arithmetic (K1n: f64 squared error, penalty "
                         "fl32(2 beta sigma^2) * length, ipynb:429-443): one K1n + K2 pass per step")
    ap.add_argument("--full-record", default=os.environ.get("VBQ_BENCH_FULL_RECORD", os.path.join(ROOT, "bench_full.json")),
                    help="where the full record goes (R-D curve arrays, per-workload and per-rank reports); stdout carries one "
                         "compact line only")
    args = ap.parse_args()
    strong = args.scaling == "strong"
    if args.workload is None:
        args.workload = "synthetic_1e8" if strong else "kodak24_c256"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # started by hand: be the launcher (no GPU call made so far)
        return launch_ranks(args.gpus)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running the {world} ranks that were launched", file=sys.stderr)
    if os.environ.get("VBQ_BENCH_TEST_STALL_RANK") == str(rank):      # testing switch: a rank stuck where a rendezvous would leave it
        time.sleep(600)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device")
    # VBQ_BENCH_ONE_DEVICE=1 (+ VBQ_BENCH_BACKEND=gloo) lets the N > 1 code path be exercised on a
    # single-GPU box: every rank uses cuda:0 and the collective runs over gloo.  Testing only.
    if os.environ.get("VBQ_BENCH_ONE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("VBQ_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # a second communicator for the small, latency-critical all-reduce of the bit-length histogram (see pipeline.py)
    args.level_group = dist.new_group() if world > 1 else None
    # ... and a gloo group to WAIT on without spinning (the ranks that idle while rank 0 times the CPU baseline)
    args.cpu_group = dist.new_group(backend="gloo", timeout=__import__("datetime").timedelta(minutes=30)) if world > 1 else None

    if args.notebook:
        out = run_notebook(args, torch, dev, cpu=not args.no_cpu_baseline)
        emit(out, args.full_record)
        return

    res = run_workload(args.workload, args, torch, dist, dev, rank, world, args.steps, args.warmup,
                       detailed=(world == 1 or not args.no_parity), strong=strong)
    out = None
    if rank == 0:
        host = res.pop("_host", None)
        out = {
            "metric": "quantized latents/sec (32-lambda sweep; one count per (element, lambda) pair and entropy-model build)",
            "value": res["value"],
            "unit": "latents/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": res["config"],
            "roofline": res["roofline"],
            "pairs_per_step": res["pairs_per_step"],
            "timed_regions_ms_per_step": res["timed_regions_ms_per_step"],
            "solves_per_s_counting_both_passes": res["solves_per_s_counting_both_passes"],
            "stages_ms": res["stages_ms"],
            "roofline_k1h": res["roofline_k1h"],
            "roofline_k2_histogram": res["roofline_k2_histogram"],
        }
        for k in ("allreduce", "per_gpu", "rd_curve"):
            if k in res:
                out[k] = res[k]
        if "parity" in res:
            out["parity"] = res["parity"]
            out["parity_vs_oracle_on_sample"] = res["parity_ok"]
        if not args.no_cpu_baseline and host is not None:
            # north_star: "next to the repo's CPU path timed on the same box's host cores in the same run" -- at EVERY N.  Rank 0
            # runs the bounded sample on its own rows after the timed regions; with N > 1 the other ranks sleep in a socket wait
            # meanwhile (a gloo barrier below: no spinning on the cores the sample uses).
            mu_h, sg_h, tab_h, ll_h, idx = host
            C = mu_h.shape[1]
            cb, idx_cpu, n = cpu_baseline(mu_h, sg_h, tab_h, ll_h, LAMBDAS)
            if world > 1:
                cb["sample"] += f"; rank 0 of {world} on its own rows, the other ranks idle"
            out["cpu_baseline"] = cb
            if idx is not None:
                # the sample doubles as a second in-run parity check of the timed configuration (pass 2)
                got = idx[:, :, :n].permute(0, 2, 1).cpu().numpy()
                out["parity_vs_oracle_on_sample"] = bool(out.get("parity_vs_oracle_on_sample", True) and np.array_equal(got, idx_cpu))
            if world == 1:
                # the same formulation in NumPy, as the reference runs it (reported next to the C port, never the target)
                out["cpu_baseline_numpy"] = cpu_baseline_numpy(mu_h, sg_h, tab_h, ll_h, LAMBDAS, idx_cpu if C > 1 else None)
        else:
            out["cpu_baseline"] = None
        del host
    del res
    torch.cuda.empty_cache()
    if world > 1 and not args.no_cpu_baseline:
        dist.barrier(group=args.cpu_group)          # the other ranks wait here (blocked on a socket) while rank 0 times the CPU sample

    # the other BASELINE configurations, same step, same checks, fewer repetitions (every rank takes part when N > 1)
    if not args.no_other_workloads:
        others = {}
        few = max(3, min(args.steps, 5))
        if world == 1:
            plan = [(w, w, LAMBDAS, False) for w in OTHER_WORKLOADS if w != args.workload]
            plan.append(("kodak24_c256_L16", "kodak24_c256", LAMBDAS_16, False))
        else:                                                    # configs[4]'s shard per rank; configs[3] split over the ranks
            plan = [("shard_1.25e8", "shard_1.25e8", LAMBDAS, False), ("synthetic_1e8_strong", "synthetic_1e8", LAMBDAS, True)]
            plan = [p for p in plan if not (p[1] == args.workload and p[3] == strong)]
        for key, w, lams, st in plan:
            r = run_workload(w, args, torch, dist, dev, rank, world, steps=few, warmup=2,
                             detailed=(world == 1 or not args.no_parity), lambdas=lams, strong=st)
            if rank == 0:
                r.pop("_host", None)
                others[key] = {"ms_per_step": r["ms_per_step"], "value": r["value"], "unit": "latents/s",
                               "scaling": ("strong" if st else "weak") if world > 1 else None,
                               "roofline": {k: r["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "pass_ms", "launches_per_pass")},
                               "stages_ms": r["stages_ms"], "k1h_latents_per_s": r["roofline_k1h"]["latents_per_s_kernel_only"],
                               "k2_frac": r["roofline_k2_histogram"]["frac"], "elements_per_gpu": r["config"]["elements_per_gpu"],
                               "workload": r["config"]["workload"], "launch": r["config"]["launch"],
                               "length_table": r["config"]["length_table"], "models": r["config"]["models"],
                               "parity": r.get("parity"), "parity_ok": r.get("parity_ok"), "allreduce": r.get("allreduce"),
                               "per_gpu": r.get("per_gpu"),
                               "rd_lagrangian_max_rel_diff_vs_oracle": (r.get("rd_curve") or {}).get("vs_oracle_on_sample", {}).get("max_rel_diff")}
            torch.cuda.empty_cache()
        if rank == 0 and world == 1:
            nb = run_notebook(args, torch, dev, workload="embeddings_1e7", steps=5, warmup=2, cpu=False)
            others["embeddings_1e7_notebook"] = {k: nb[k] for k in ("ms_per_step", "value", "unit", "roofline", "parity_vs_oracle_on_sample")}
            others["embeddings_1e7_notebook"]["workload"] = nb["config"]["workload"]
            others.update(run_call_patterns(torch, dev))
            others.update(run_next_rows(torch, dev))
        if rank == 0:
            out["workloads"] = others
    if rank == 0:
        emit(out, args.full_record)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        bad = [k for k, v in out.get("workloads", {}).items() if v.get("parity_ok", v.get("parity_vs_oracle_on_sample")) is False]
        if out.get("parity_vs_oracle_on_sample") is False or bad:
            raise SystemExit(f"parity against the oracle FAILED (headline: {out.get('parity_vs_oracle_on_sample')}; workloads: {bad}); "
                             "the R-D Lagrangian gate is 1e-5 relative, indices must be identical")


def run_call_patterns(torch, dev, steps=10, warmup=3):
    """The calls the reference's own code makes, under the same clock as the headline: the literal quantize(mu, sigma, lmbda)
    on channel-last latents (layout change inside the events), the raw-length sweep on planes, one lambda per call, and the
    notebook's one-beta / fifty-beta calls.  Every line: its own algorithmic bytes (8 + 2 L) E and oracle parity."""
    import vbq_amd
    from vbq_amd import embeddings as Emb, ops
    from oracle import c_oracle as CO, vbq_oracle as O
    out = {}
    rows, C, desc = WORKLOADS["kodak24_c256"]
    E = rows * C
    mu_h, sg_h = make_inputs(rows, C, seed=1000)
    mu_bc, sg_bc = torch.from_numpy(mu_h).to(dev), torch.from_numpy(sg_h).to(dev)
    tab_h = empirical_tables(mu_bc, rows, C, "bc")
    tab = torch.from_numpy(tab_h).to(dev)
    mu, sg = ops.transpose(mu_bc), ops.transpose(sg_bc)
    th = CO.max_threads()
    n = 2048                                                   # parity: the first rows of every channel against the C oracle
    want = {}

    def oracle(lams):
        key = tuple(lams)
        if key not in want:
            want[key] = CO.quantize(mu_h[:n], sg_h[:n], tab_h, list(lams), N=N_BITS, threads=th)       # [L, n, C]
        return want[key]

    def line(ms, L, kernel, what, ok, extra=None):
        alg = E * (8 + 2 * L)
        d = {"ms_per_step": ms, "value": E * L / (ms * 1e-3), "unit": "latents/s",
             "roofline": {"bound": "hbm", "limited_by": "valu issue", "kernel": kernel, "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                          "frac": alg / (ms * 1e-3) / HBM_PEAK, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms},
             "parity_vs_oracle_on_sample": ok, "workload": f"kodak24_c256: {desc}; {what}"}
        d.update(extra or {})
        return d

    # (1) raw-length sweep on planes: quantize() over the 32-point sweep, indices out, no histogram (K1e)
    L = len(LAMBDAS)
    idx = torch.empty((L, C, rows), dtype=torch.uint16, device=dev)
    ms = event_ms(torch, lambda: ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb", out_idx=idx), steps, warmup)
    ok = bool(np.array_equal(idx[:, :, :n].cpu().numpy().transpose(0, 2, 1), oracle(LAMBDAS)))
    out["kodak24_c256_quantize_sweep_raw"] = line(ms, L, "k_quant_hull_idx", f"one pass of quantize(mu, sigma, lmbda) over the {L}-point "
                                                  "sweep, raw code lengths, channel-major planes in, rank indices out", ok)
    planes_ms = ms
    del idx
    # (2) the literal call on latents as they arrive: channel-last in, channel-last out, layout changes INSIDE the events
    res = {}
    def facade():
        res["idx"] = vbq_amd.quantize(mu_bc, sg_bc, LAMBDAS, table=tab)
    ms = event_ms(torch, facade, steps, warmup)
    ok = bool(np.array_equal(res["idx"][:, :n].cpu().numpy(), oracle(LAMBDAS)))
    tr_in = event_ms(torch, lambda: (ops.transpose(mu_bc), ops.transpose(sg_bc)), steps, 2)
    idx_p = torch.empty((L, C, rows), dtype=torch.uint16, device=dev)
    tr_out = event_ms(torch, lambda: ops.transpose_planes(idx_p, out=res["idx"]), steps, 2)
    del idx_p
    res.clear()
    out["facade_bc_L32"] = line(ms, L, "vbq_transpose_f32 x2 + k_quant_hull_idx + vbq_transpose_planes",
                                "vbq_amd.quantize(mu[B, C], sigma[B, C], lmbda[32], table): channel-last device tensors in, "
                                "channel-last indices out; both layout changes inside the events", ok,
                                {"planes_kernel_ms": planes_ms, "input_transposes_ms": tr_in, "output_transpose_ms": tr_out,
                                 "facade_over_planes_plus_transposes": ms / (planes_ms + tr_in + tr_out)})
    torch.cuda.empty_cache()
    # (2b) the same call with the results left as the kernels write them: no transpose back (out_layout="planes")
    def facade_planes():
        res["idx"] = vbq_amd.quantize(mu_bc, sg_bc, LAMBDAS, table=tab, out_layout="planes")
    ms = event_ms(torch, facade_planes, steps, warmup)
    ok = bool(np.array_equal(res["idx"][:, :, :n].cpu().numpy().transpose(0, 2, 1), oracle(LAMBDAS)))
    prep_ms = event_ms(torch, lambda: ops.prep_planes(mu_bc, sg_bc), steps, 2)
    res.clear()
    out["facade_bc_L32_planes_out"] = line(ms, L, "vbq_prep_planes_f32 + k_quant_hull_idx",
                                           "vbq_amd.quantize(mu[B, C], sigma[B, C], lmbda[32], table, out_layout='planes'): "
                                           "channel-last in, index planes [L, C, B] out", ok,
                                           {"planes_kernel_ms": planes_ms, "prep_planes_ms": prep_ms,
                                            "facade_over_planes_plus_prep": ms / (planes_ms + prep_ms)})
    # (2c) Z_hat alone, channel-last (return_values=True, return_indices=False): solve on planes + ONE pass that looks the values
    #      up and changes the layout (vbq_gather_latents_u16)
    def facade_values():
        res["z"] = vbq_amd.quantize(mu_bc, sg_bc, LAMBDAS, table=tab, return_values=True, return_indices=False)
    ms = event_ms(torch, facade_values, steps, warmup)
    srt_h = np.sort(tab_h, axis=1)
    ok = bool(np.array_equal(res["z"][:, :n].cpu().numpy(), srt_h[np.arange(C)[None, None, :], oracle(LAMBDAS).astype(np.int64)]))
    res.clear()
    out["facade_bc_L32_values_out"] = line(ms, L, "vbq_prep_planes_f32 + k_quant_hull_idx + k_gather_latents",
                                           "vbq_amd.quantize(..., return_values=True, return_indices=False): channel-last in, "
                                           "Z_hat channel-last out (f32: 4 B per latent written instead of 2)", ok)
    torch.cuda.empty_cache()
    # (3) the sweep of post_process.py:115 through the same call (16 lambdas, raw lengths)
    L16 = len(LAMBDAS_16)
    idx = torch.empty((L16, C, rows), dtype=torch.uint16, device=dev)
    ms = event_ms(torch, lambda: ops.quantize(mu, sg, tab, LAMBDAS_16, N=N_BITS, layout="cb", out_idx=idx), steps, warmup)
    ok = bool(np.array_equal(idx[:, :, :n].cpu().numpy().transpose(0, 2, 1), oracle(LAMBDAS_16)))
    out["kodak24_c256_quantize_sweep_raw_L16"] = line(ms, L16, "k_quant_hull_idx", "one pass of quantize() over 2**linspace(-8,7,16) "
                                                      "(post_process.py:115), raw code lengths, planes in, indices out", ok)
    # (4) ONE lambda per call (the north star's literal signature): smallest, middle and largest lambda of the sweep
    one = []
    oks = True
    for l in (0, 8, 16, 24, 31):
        lam = [LAMBDAS[l]]
        ms1 = event_ms(torch, lambda: ops.quantize(mu, sg, tab, lam, N=N_BITS, layout="cb", out_idx=idx[:1]), steps, warmup)
        oks = oks and bool(np.array_equal(idx[:1, :, :n].cpu().numpy().transpose(0, 2, 1), oracle(LAMBDAS)[l:l + 1]))
        one.append({"lambda": LAMBDAS[l], "ms": ms1, "latents_per_s": E / (ms1 * 1e-3)})
    med = float(np.median([o["ms"] for o in one]))
    out["kodak24_c256_L1"] = line(med, 1, "k_quant_pruned / k_quant_fast (one lambda)", "quantize(mu, sigma, lmbda) with ONE lambda per call, "
                                  "planes in, indices out; median over five lambdas of the sweep", oks, {"per_lambda": one})
    del idx
    torch.cuda.empty_cache()
    out.update(run_api_methods(torch, dev, mu_h, sg_h, mu_bc, sg_bc, tab_h, steps=steps, warmup=warmup))
    del mu, sg, mu_bc, sg_bc
    torch.cuda.empty_cache()
    # (5) the notebook's calls: compress_coordinates(means, stds, beta) with one beta (ipynb:466) and the 50-beta sweep of
    #     cell 32 (ipynb:1102) in one launch; 100000 x 100 embeddings
    rows1 = WORKLOADS["embeddings_1e7"][0]
    m_h, s_h = make_inputs(rows1, 1, seed=1000)
    m, s = torch.from_numpy(m_h.reshape(rows1)).to(dev), torch.from_numpy(s_h.reshape(rows1)).to(dev)
    pts_h, lens_h = Emb.make_code_book(Emb.empirical_std(m), N_BITS)                  # ipynb:373-390
    cb = torch.from_numpy(pts_h).to(dev)
    r2s = O.level_major_to_rank(N_BITS)
    nn = 20000
    for key, betas, what in (("embeddings_1e7_beta1", [BETAS_50[25]], "compress_coordinates(means, stds, beta) with ONE beta per call (ipynb:466)"),
                             ("embeddings_1e7_beta50", BETAS_50, "the notebook's 50-beta sweep exp(linspace(log 0.01, log 1e5, 50)) (ipynb cell 32) in one launch")):
        Lb = len(betas)
        ix = torch.empty((Lb, rows1), dtype=torch.uint16, device=dev)
        ms = event_ms(torch, lambda: ops.quantize_notebook(m, s, cb, betas, N=N_BITS, want_values=False, out_idx=ix), 5, 2)
        got = ix[:, :nn].cpu().numpy().astype(np.int64)
        check = list(range(Lb)) if Lb <= 12 else sorted(set(list(range(0, Lb, 5)) + [Lb - 1]))     # every fifth beta of a long sweep
        ok = all(np.array_equal(got[i], r2s[CO.compress_coordinates(m_h[:nn, 0], s_h[:nn, 0], betas[i], pts_h, lens_h, threads=th)[1]])
                 for i in check)
        alg = rows1 * (8 + 2 * Lb)
        out[key] = {"ms_per_step": ms, "value": rows1 * Lb / (ms * 1e-3), "unit": "latents/s",
                    "roofline": {"bound": "hbm", "limited_by": "valu issue", "kernel": "k_quant_notebook_hull" if Lb >= 6 else "k_quant_notebook_fast",
                                 "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg / (ms * 1e-3) / HBM_PEAK,
                                 "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms},
                    "parity_vs_oracle_on_sample": bool(ok),
                    "workload": f"embeddings_1e7: {WORKLOADS['embeddings_1e7'][2]}; {what}; notebook arithmetic (f64 squared error, ipynb:429-443), indices out"}
        del ix
    return out


MFMA_F32_PEAK = 157.3e12      # MI355X dense f32 matrix peak (MI355X_MICROARCH.md)
VALU_F32_PEAK = 157.3e12      # f32 vector peak (256 CUs x 128 FMA lanes x 2 x 2.4 GHz)


def run_next_rows(torch, dev, steps=5, warmup=2):
    """The rows of SURVEY 8(f) and the prior's CDF (K4) under the same clock as the headline -- each with the roofline that bounds
    it and a parity verdict against its checker (outside the timed region):
      rans_encode / rans_decode    the entropy coder on the index planes of the Kodak-24 sweep (quantizer.py:144,226-228 made real):
                                   3.0e8 symbols in 8192 streams; HBM bound on 2 B per symbol + the compressed words
      prediction_ranks_1e5x100     the analogy evaluator (ipynb:199-209): 19544 questions x 100000 words x 100 dims, fused f32-MFMA
                                   GEMM + rank count; bound "mfma" against the dense f32 matrix peak
      ms_ssim_kodak                MS-SSIM (img_comparison_metrics.py:160) of one 512 x 768 RGB image against 16 reconstructions
      bmshj_icdf_table_c256        BMSHJ2018Prior.inverse_cdf on the [2047, 256] xi grid (learned_prior.py:173-218): the code-point
                                   table of build_code_points; f32 vector arithmetic (parity: NumPy restatement, UNPINNED)
      bmshj_fit_pass               one NLL + gradient pass of the prior fit (learned_prior.py:363-462) over 500 x 1536 x 256 latents"""
    from vbq_amd import embeddings as Emb, metrics as Mx, ops, priors
    from vbq_amd.coder import RansCodec, ideal_bits, quantize_frequencies
    from oracle import c_oracle as CO, vbq_oracle as O
    out = {}
    th = CO.max_threads()

    def entry(ms, bound, achieved, peak, roof_unit, kernel, ok, what, **extra):
        d = {"ms_per_step": ms, "roofline": {"bound": bound, "kernel": kernel, "achieved": achieved, "peak": peak, "unit": roof_unit,
                                             "frac": achieved / peak, "avg_launch_ms": ms},
             "parity_vs_oracle_on_sample": bool(ok), "workload": what}
        d.update(extra)
        return d

    # ---- rANS on the Kodak-24 sweep's indices
    rows, C, _ = WORKLOADS["kodak24_c256"]
    L = len(LAMBDAS)
    mu_h, sg_h, tab_h = make_inputs_with_table(rows, C, 1000)
    mu, sg = (torch.from_numpy(np.ascontiguousarray(a.T)).to(dev) for a in (mu_h, sg_h))
    idx = ops.quantize(mu, sg, torch.from_numpy(tab_h).to(dev), LAMBDAS, N=N_BITS, layout="cb")            # [L, C, rows]
    del mu, sg
    counts = ops.histogram(idx, C, N=N_BITS, layout="cb")
    freq = quantize_frequencies(counts)
    codec = RansCodec(freq.reshape(-1, T), N=N_BITS)
    res = {}
    def enc():
        res["w"], res["s"] = codec.encode(idx)
    ms_e = event_ms(torch, enc, steps, warmup)
    words, sizes = res["w"], res["s"]
    def dec():
        res["back"] = codec.decode(words, sizes, rows)
    ms_d = event_ms(torch, dec, steps, warmup)
    nsym = idx.numel()
    cbytes = codec.compressed_bits(sizes) / 8
    ok = bool(torch.equal(res["back"].view(torch.int16).reshape(-1), idx.view(torch.int16).reshape(-1)))
    ns = 8                                                    # the first streams against the C checker, word for word
    i_h = idx.reshape(-1, rows)[:ns].cpu().numpy()
    f_h = np.asarray(freq).reshape(-1, T)[:ns]
    w_ref, s_ref = CO.rans_encode(i_h, f_h, codec.segment)
    s_got = sizes.reshape(-1, sizes.shape[-1])[:ns].cpu().numpy()
    keep = np.arange(codec.segment + 2)[None, None, :] < s_ref[..., None].astype(np.int64)
    ok = ok and bool(np.array_equal(s_got, s_ref)) and \
        bool(np.array_equal(words.reshape(-1, words.shape[-2], words.shape[-1])[:ns].cpu().numpy()[keep], w_ref[keep]))
    est = ideal_bits(counts, freq)
    ok = ok and est <= 8 * cbytes <= 1.02 * est
    rate = {"bits_per_symbol": 8 * cbytes / nsym, "cross_entropy_bits_per_symbol": est / nsym, "symbols": nsym, "streams": L * C}
    for key, ms, kern in (("rans_encode", ms_e, "k_rans_encode"), ("rans_decode", ms_d, "k_rans_decode")):
        alg = 2.0 * nsym + cbytes
        out[key] = entry(ms, "hbm", alg / (ms * 1e-3) / 1e9, HBM_PEAK / 1e9, "GB/s", kern, ok,
                         f"rANS {key[5:]} of the kodak24_c256 sweep's index planes: {nsym:.3g} symbols in {L * C} streams, segments of "
                         f"{codec.segment}; decode(encode) == indices, first {ns} streams word for word == the C checker, size within 2 % of "
                         "the cross-entropy", value=nsym / (ms * 1e-3), unit="symbols/s", algorithmic_bytes_per_launch=alg, **rate)
    del idx, words, sizes, res, counts
    torch.cuda.empty_cache()

    # ---- analogy evaluator
    V, K, Q = 100_000, 100, 19_544
    rng = np.random.default_rng(0)
    emb_h = rng.normal(0, 1, (V, K)).astype(np.float32)
    an = rng.integers(0, V, (Q, 4)).astype(np.int32)
    emb = torch.from_numpy(emb_h).to(dev)
    got = {}
    def ranks():
        got["r"] = Emb.prediction_ranks(emb, an)
    ms = event_ms(torch, ranks, steps, warmup)
    nq = 256
    ok = bool(np.array_equal(got["r"][:nq].cpu().numpy(), CO.analogy_ranks(emb_h, an[:nq], threads=th)))
    flop = 2.0 * Q * V * K
    out["prediction_ranks_1e5x100"] = entry(ms, "mfma", flop / (ms * 1e-3) / 1e12, MFMA_F32_PEAK / 1e12, "TFLOP/s", "k_rank_gemm", ok,
                                            f"prediction_ranks (ipynb:199-209): {Q} questions x {V} words x {K} dims, f32 MFMA; first {nq} "
                                            "ranks == the C checker (same fma chain)", value=Q / (ms * 1e-3), unit="questions/s",
                                            algorithmic_flops_per_launch=flop)
    del emb, got
    # ---- MS-SSIM, one Kodak-sized image against 16 reconstructions (device-resident uint8)
    H, W, M = 512, 768, 16
    yy, xx = np.mgrid[0:H, 0:W]
    x = np.clip(128 + 80 * np.sin(yy / 17.0)[..., None] * np.cos(xx / 11.0)[..., None] + rng.normal(0, 10, (H, W, 3)), 0, 255).astype(np.uint8)
    xs = np.repeat(x[None], M, axis=0)
    ys = np.clip(xs + rng.normal(0, 1, xs.shape) * (1 + 2 * np.arange(M))[:, None, None, None], 0, 255).astype(np.uint8)
    xd, yd = torch.from_numpy(xs).to(dev), torch.from_numpy(ys).to(dev)
    def ssim():
        got["s"] = Mx.ms_ssim(xd, yd)
    got = {}
    ms = event_ms(torch, ssim, steps, warmup)
    pick = [0, M - 1]
    ok = bool(np.allclose(got["s"][pick], O.ms_ssim(xs[pick], ys[pick]), rtol=1e-12, atol=0))
    alg = sum(2.0 * M * ((H + (1 << i) - 1) >> i) * ((W + (1 << i) - 1) >> i) * 3 * 8 for i in range(5))
    out["ms_ssim_kodak"] = entry(ms, "hbm", alg / (ms * 1e-3) / 1e9, HBM_PEAK / 1e9, "GB/s", "k_ssim_*", ok,
                                 f"ms_ssim (img_comparison_metrics.py:160) of one {H} x {W} RGB image against {M} reconstructions, uint8 on the "
                                 "device, float64 arithmetic; algorithmic bytes = both f64 images read once per scale; two pairs == the "
                                 "NumPy restatement to 1e-12", value=M / (ms * 1e-3), unit="image pairs/s", algorithmic_bytes_per_launch=alg)
    del xd, yd
    # ---- BMSHJ2018: the code-point table (inverse CDF by bisection) and one pass of the fit
    Cp = 256
    p = priors.BMSHJ2018Prior(Cp, init_scale=10.0, seed=0)
    xi = np.repeat(XI[:, None], Cp, axis=1).astype(np.float32)
    def icdf():
        got["t"] = p.inverse_cdf(xi)
    ms = event_ms(torch, icdf, 3, 1)
    cs = 16                                                  # the reference's stopping rule is global: the checker gets its own small call
    ps = priors.BMSHJ2018Prior(cs, init_scale=10.0, seed=0)
    orc = O.BMSHJ2018Oracle(*O.BMSHJ2018Oracle.effective(ps.matrices, ps.biases, ps.factors))
    want = orc.inverse_cdf(xi[:, :cs])
    ok = bool(np.allclose(np.asarray(ps.inverse_cdf(xi[:, :cs])), want, rtol=1e-4, atol=1e-5))
    tab = np.asarray(got["t"])
    ok = ok and bool(np.all(np.isfinite(tab))) and tab.shape == xi.shape
    its = (getattr(p, "last_iterations", None) or 0) + 1
    flop = 120.0 * xi.size * (its or 40)
    out["bmshj_icdf_table_c256"] = entry(ms, "valu", flop / (ms * 1e-3) / 1e12, VALU_F32_PEAK / 1e12, "TFLOP/s", "k_bmshj_icdf_step", ok,
                                         f"BMSHJ2018Prior.inverse_cdf on the [{T}, {Cp}] xi grid (learned_prior.py:173-218), the table of "
                                         f"build_code_points; ~120 flop per point and bisection step, {its or '~40'} steps enqueued as one chain (the stopping "
                                         f"rule applied on the device); parity UNPINNED "
                                         f"(no TensorFlow): a {cs}-channel call against the NumPy restatement to 1e-4",
                                         value=xi.size / (ms * 1e-3), unit="code points/s", parity_pinned=False)
    n_fit = 500 * 1536
    scale = np.exp(rng.uniform(np.log(0.3), np.log(3.0), Cp)).astype(np.float32)
    x_cb = torch.from_numpy(scale[:, None] * rng.standard_normal((Cp, n_fit), dtype=np.float32)).to(dev)
    params = p._params()
    acc = torch.zeros((Cp, 44), dtype=torch.float64, device=dev)
    def fit_pass():
        acc.zero_()
        ops.bmshj_nll_grad(params, x_cb, out=acc)
    ms = event_ms(torch, fit_pass, steps, warmup)
    nsub = 512
    sub = x_cb[:, :nsub].contiguous()
    g = ops.bmshj_nll_grad(params, sub).cpu().numpy()
    orc = O.BMSHJ2018Oracle(*O.BMSHJ2018Oracle.effective(p.matrices, p.biases, p.factors))
    want = -np.sum(orc.logpdf(sub.cpu().numpy().T).astype(np.float64), axis=0)              # per channel
    ok = bool(np.allclose(g[:, 43], want, rtol=1e-4))
    Ef = float(Cp) * n_fit
    out["bmshj_fit_pass"] = entry(ms, "hbm", 4.0 * Ef / (ms * 1e-3) / 1e9, HBM_PEAK / 1e9, "GB/s", "k_bmshj_nll_grad", ok,
                                  f"one NLL + gradient pass of BMSHJ2018Prior.fit (learned_prior.py:363-462) over 500 x 1536 x {Cp} latents "
                                  f"(post_process.py:68-81), 4 B read per element; parity UNPINNED: the loss of {nsub} rows per channel against "
                                  "the NumPy restatement to 1e-4", value=Ef / (ms * 1e-3), unit="elements/s",
                                  algorithmic_bytes_per_launch=4.0 * Ef, parity_pinned=False)
    del x_cb, acc
    torch.cuda.empty_cache()
    return out


def run_api_methods(torch, dev, mu_h, sg_h, mu_bc, sg_bc, tab_h, steps=10, warmup=3):
    """The reference-facing METHODS under the clock (not the pipeline object behind them):
      build_entropy_models_api   ChannelwisePriorCDFQuantizer.build_entropy_models_from_latents end to end on the Kodak-24
                                 tensor (what post_process.py:103 triggers after the encoder), channel-last device tensors in,
                                 nothing read on the host: K calls back to back, one synchronisation at the end;
      compress_latents_image     the per-image call of the evaluation loop (utils.py:542 -> quantizer.py:190-240): one image
                                 [1, 32, 48, 256], the 16 lambdas of post_process.py:115, corrected lengths, results left on
                                 the device (return_np=False) and as NumPy arrays (the reference's form; PCIe-bound).
    Parity: the three tables against a second build through the pipeline object and NumPy's -log2; the image's Z_hat /
    raw_num_bits / num_bits against the C oracle's indices pushed through the same tables."""
    import vbq_amd
    from vbq_amd import entropy, ops
    from vbq_amd.pipeline import EntropyModelBuild
    from oracle import c_oracle as CO, vbq_oracle as O
    out = {}
    rows, C = mu_h.shape
    E = rows * C

    class _Table:                                             # prior stand-in: inverse_cdf returns the bench's code book
        def inverse_cdf(self, xi):
            return np.ascontiguousarray(tab_h.T)
    q = vbq_amd.ChannelwisePriorCDFQuantizer(C, N_BITS)
    q.build_code_points(_Table())
    for key, lams in (("build_entropy_models_api", LAMBDAS), ("build_entropy_models_api_L16", LAMBDAS_16)):
        L = len(lams)

        def build_call():
            q.build_entropy_models_from_latents(mu_bc, sg_bc, lams, 1)
        for _ in range(warmup):
            build_call()
        clock_ramp(torch, build_call, min(RAMP_S, 0.15))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            build_call()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        on_dev = bool(q.entropy_models.on_device and q.raw_code_length_entropy_models.on_device)
        # the pipeline object on the same planes (what the headline times), for the ratio and as the second route
        mu_p, sg_p = ops.prep_planes(mu_bc, sg_bc)
        build = EntropyModelBuild(rows, C, lams, torch.from_numpy(tab_h).to(dev), N=N_BITS, add_n_smoothing=1)
        pipe_ms = event_ms(torch, lambda: (ops.prep_planes(mu_bc, sg_bc, out_mu=mu_p, out_sigma=sg_p), build.run(mu_p, sg_p)), steps, 2)
        torch.cuda.synchronize()
        got_m = np.stack([q.entropy_models[l] for l in lams])                     # first host read (one copy + deferred checks)
        got_r = np.stack([q.raw_code_length_entropy_models[l] for l in lams])
        got_c = np.stack([q._code_counts[l] for l in lams])
        ok = bool(np.array_equal(got_m, build.models.cpu().numpy()) and np.array_equal(got_r, build.raw_models.cpu().numpy())
                  and np.array_equal(got_c, build.counts.cpu().numpy())
                  and np.array_equal(got_m, entropy.neg_log2_freq(got_c, 1)) and int(got_c.sum()) == E * L)
        alg = E * (8 + 2 * L)
        out[key] = {"ms_per_step": ms, "value": E * L / (ms * 1e-3), "unit": "latents/s",
                    "roofline": {"bound": "hbm", "limited_by": "valu issue", "kernel": "k_quant_fast (inside the method)",
                                 "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                 # the whole call against ONE pass's algorithmic bytes: a lower bound of the kernel's fraction
                                 "frac": alg / (ms * 1e-3) / HBM_PEAK, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms},
                    "pipeline_step_ms_same_inputs": pipe_ms, "api_over_pipeline_step": ms / pipe_ms,
                    "tables_device_resident_after_the_calls": on_dev, "parity_vs_oracle_on_sample": ok,
                    "workload": f"kodak24_c256 [{rows} x {C}]: ChannelwisePriorCDFQuantizer.build_entropy_models_from_latents(means, stds, "
                                f"{L} lambdas, add_n_smoothing=1), channel-last device tensors in, no host read; {steps} calls, one synchronisation"}
        del build, mu_p, sg_p
        torch.cuda.empty_cache()
    # ---- one image through compress_latents with the 16-lambda models built above
    lams = LAMBDAS_16
    L = len(lams)
    H, W = 32, 48
    B = H * W
    m_img = mu_bc[:B].reshape(1, H, W, C).contiguous()
    lv_img = (2.0 * torch.log(sg_bc[:B])).reshape(1, H, W, C).contiguous()
    res = {}

    def img_dev():
        res["o"] = q.compress_latents(m_img, lv_img, lams, return_np=False)
    for _ in range(warmup):
        img_dev()
    clock_ramp(torch, img_dev, min(RAMP_S, 0.15))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5 * steps):
        img_dev()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (5 * steps) * 1e3
    t0 = time.perf_counter()
    for _ in range(steps):
        img_dev()
        torch.cuda.synchronize()
    ms_sync = (time.perf_counter() - t0) / steps * 1e3
    # its kernels one by one (events; each repeated alone): planes (with sigma = sqrt(exp(logvar))), solve, fused lookups
    ll_d, md_d = q._level_len_dev(lams), q._models_dev(lams)
    tab_d, srt_d = q._table_dev(), q._sorted_dev()
    mu_p, sg_p = ops.prep_planes(m_img.reshape(B, C), lv_img.reshape(B, C), spread="logvar")
    idx_p = torch.empty((L, C, B), dtype=torch.uint16, device=dev)
    k_prep = event_ms(torch, lambda: ops.prep_planes(m_img.reshape(B, C), lv_img.reshape(B, C), spread="logvar", out_mu=mu_p, out_sigma=sg_p), 50, 5)
    k_solve = event_ms(torch, lambda: ops.quantize(mu_p, sg_p, tab_d, lams, N=N_BITS, level_len=ll_d, layout="cb", out_idx=idx_p), 50, 5)
    k_gather = event_ms(torch, lambda: ops.gather_latents(idx_p, N=N_BITS, table_sorted=srt_d, level_len=ll_d, models=md_d, want_num_bits=True), 50, 5)
    kernels = k_prep + k_solve + k_gather
    # parity: the oracle's indices for this image (sigma as the device derived it) pushed through the same tables
    sg_img_h = sg_p.t().contiguous().cpu().numpy()
    ll_h = ll_d.cpu().numpy()
    wi = CO.quantize(mu_h[:B], sg_img_h, tab_h, lams, N=N_BITS, level_len=ll_h, threads=CO.max_threads()).astype(np.int64)   # [L, B, C]
    srt_h = np.sort(tab_h, axis=1)
    ch = np.arange(C)[None, None, :]
    ls = np.arange(L)[:, None, None]
    lev = O.levels_of_sorted_ranks(N_BITS)[wi]
    md_h = md_d.cpu().numpy()
    o = res["o"]
    ok = all(np.array_equal(o["Z_hat"][l].cpu().numpy().reshape(B, C), srt_h[ch[0], wi[i]]) and
             np.array_equal(o["raw_num_bits"][l].cpu().numpy().reshape(B, C), ll_h[i][ch[0], lev[i]]) and
             np.array_equal(o["num_bits"][l].cpu().numpy().reshape(B, C), md_h[i][ch[0], wi[i]]) for i, l in enumerate(lams))
    alg = B * C * (8 + 12 * L)                                   # 8 B in per element, three f32 results per (element, lambda)
    out["compress_latents_image"] = {
        "ms_per_step": ms, "value": B * C * L / (ms * 1e-3), "unit": "latents/s",
        "roofline": {"bound": "hbm", "kernel": "k_prep_planes + k_quant_fast + k_gather_latents", "achieved": alg / (ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg / (ms * 1e-3) / HBM_PEAK, "algorithmic_bytes_per_launch": alg,
                     "avg_launch_ms": ms},
        "ms_per_call_synchronised": ms_sync, "kernels_ms": {"prep_planes": k_prep, "solve": k_solve, "gather_latents": k_gather},
        "sum_of_kernels_ms": kernels, "call_over_sum_of_kernels": ms / kernels, "launches_per_call": 3,
        "parity_vs_oracle_on_sample": bool(ok),
        "workload": f"one Kodak image [1, {H}, {W}, {C}]: ChannelwisePriorCDFQuantizer.compress_latents(means, logvars, {L} lambdas of "
                    f"post_process.py:115, return_np=False) with corrected lengths and entropy models on the device; {5 * steps} calls, one synchronisation"}
    # the reference's own form: NumPy in, NumPy out and EVERYTHING READ on the host (PCIe-bound: 3 x L x B x C x 4 B back per
    # call) -- never the headline.  return_np=True hands out lazy views (vbq_amd.lazy): a quantity crosses PCIe when it is first read
    m_np, lv_np = m_img.cpu().numpy(), lv_img.cpu().numpy()
    keys3 = ("Z_hat", "raw_num_bits", "num_bits")

    def img_numpy(read=keys3):
        o_ = q.compress_latents(m_np, lv_np, lams)
        for k in read:
            np.asarray(o_[k][lams[0]])               # first host access of the quantity: all its lambdas come over at once
        return o_
    for _ in range(2):
        o_np = img_numpy()
    t0 = time.perf_counter()
    for _ in range(steps):
        o_np = img_numpy()
    ms_np = (time.perf_counter() - t0) / steps * 1e3
    ok_np = all(isinstance(np.asarray(o_np[k][l]), np.ndarray) and np.array_equal(o_np[k][l], o[k][l].cpu().numpy()) for k in keys3 for l in lams)
    t0 = time.perf_counter()
    for _ in range(steps):
        img_numpy(("num_bits",))
    ms_np1 = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    out["compress_latents_image_numpy"] = {
        "ms_per_step": ms_np, "value": B * C * L / (ms_np * 1e-3), "unit": "latents/s",
        "roofline": {"bound": "pcie", "kernel": "host copies", "frac": None, "bytes_to_host_per_call": 3 * L * B * C * 4},
        "ms_per_step_reading_num_bits_only": ms_np1,
        "parity_vs_oracle_on_sample": bool(ok and ok_np),
        "workload": f"the same call with NumPy arrays in and out (the reference's form, return_np=True) and all three quantities read on "
                    f"the host: {3 * L * B * C * 4 / 1e6:.0f} MB to the host per call through the pinned staging blocks (reading num_bits "
                    f"alone: ms_per_step_reading_num_bits_only)"}
    # ---- the evaluation loop's use of it (utils.py:542-554): compress(X, vae, settings) with a VAE that lives on the device, then what
    # the loop reads -- np.sum(num_bits), np.sum(num_bits_cl) per setting and X_hat as uint8 -- without any latent-shaped array
    # crossing PCIe: Z_hat goes to the decoder as the device tensor the kernels wrote, the sums are taken on the device in NumPy's
    # float32 order (vbq_numpy_row_sums_f32), 2 L floats and the (stand-in) uint8 reconstructions come to the host.
    from vbq_amd import utils as vutils

    class _DeviceVAE:                                         # stand-in: the real encoder / decoder are conv nets (out of scope)
        def encode(self, X):
            return m_img, lv_img

        def decode(self, Z):
            return 0.5 + 0.1 * Z[..., :3]
    vae = _DeviceVAE()
    X = np.zeros((1, H, W, 3), np.float32)
    seen, pinned = {}, {}

    def eval_image_eager():                                  # the calls one by one (round 5's form; what a VAE that cannot be captured gets)
        tmp = q.compress(X, vae, lams, clip=True)
        seen["sums"], seen["sums_cl"], seen["x_u8"] = vutils.evaluation_reads(tmp, lams, pinned)      # one synchronisation
        seen["tmp"] = tmp

    def eval_image():                                        # the same calls as ONE HIP graph replay per image (vbq_amd.replay)
        seen["tmp"], (seen["sums"], seen["sums_cl"], seen["x_u8"]) = q.compress_replay(X, vae, lams, clip=True)

    def loop_ms(fn):
        for _ in range(warmup):
            fn()
        clock_ramp(torch, fn, min(RAMP_S, 0.15))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5 * steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (5 * steps) * 1e3
    before = q._stager().transfers
    ms_eager = loop_ms(eval_image_eager)
    eager_seen = {k: np.array(seen[k]) for k in ("sums", "sums_cl", "x_u8")}
    ms_ev = loop_ms(eval_image)
    rp_mode = next(iter(q._dev_cache.get("_replays", {}).values())).mode if q._dev_cache.get("_replays") else None
    no_copies = q._stager().transfers == before and all(np.array_equal(eager_seen[k], seen[k]) for k in eager_seen)
    md_sums = np.array([np.sum(md_h[i][ch[0], wi[i]].reshape(1, H, W, C)[0]) for i in range(L)], dtype=np.float32)       # the oracle's indices
    cl_sums = np.array([np.sum(ll_h[i][ch[0], lev[i]].reshape(1, H, W, C)[0]) for i in range(L)], dtype=np.float32)
    z_or = np.stack([srt_h[ch[0], wi[i]] for i in range(L)]).reshape(L, H, W, C)
    x_or = np.clip(np.round(np.clip(np.float32(0.5) + np.float32(0.1) * z_or[..., :3], 0, 1) * 255), 0, 255).astype(np.uint8)
    ok_ev = bool(np.array_equal(seen["sums"], md_sums) and np.array_equal(seen["sums_cl"], cl_sums) and np.array_equal(seen["x_u8"], x_or)
                 and no_copies)
    out["evaluate_loop_image"] = {
        "ms_per_step": ms_ev, "value": B * C * L / (ms_ev * 1e-3), "unit": "latents/s",
        "roofline": {"bound": "hbm", "kernel": "k_prep_planes + k_quant_fast + k_gather_latents + k_np_block_sums", "achieved": alg / (ms_ev * 1e-3) / 1e9,
                     "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg / (ms_ev * 1e-3) / HBM_PEAK, "algorithmic_bytes_per_launch": alg,
                     "avg_launch_ms": ms_ev},
        "latent_shaped_arrays_copied_to_the_host": 0 if no_copies else None, "parity_vs_oracle_on_sample": ok_ev,
        "ms_per_step_eager_launches": ms_eager, "replay_mode": rp_mode,
        "workload": f"one image of the evaluation loop (utils.py:542-554): quantizer.compress(X, vae, {L} lambdas, clip=True) with a VAE "
                    f"stand-in on the device (decode gets the device Z_hat), then np.sum(num_bits) / np.sum(num_bits_cl) per lambda on the device "
                    f"in NumPy's float32 order and X_hat as uint8: {2 * L} floats + the uint8 images reach the host; one synchronising read per "
                    f"image; {5 * steps} images; ONE HIP graph replay per image (quantizer.compress_replay: the same calls captured once per "
                    f"shape; ms_per_step_eager_launches = the calls one by one)"}
    return out


def run_notebook(args, torch, dev, workload=None, steps=None, warmup=None, cpu=True):
    """The C = 1 workloads in the notebook's own arithmetic: one K1n launch + one K2 launch per step."""
    from vbq_amd import ops, embeddings as Emb
    name = workload or (args.workload if WORKLOADS[args.workload][1] == 1 else "embeddings_1e7")
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    rows, C, desc = WORKLOADS[name]
    L = len(LAMBDAS)
    BETAS = [float(b) for b in np.exp(np.linspace(np.log(0.01), np.log(1e5), L))]      # ipynb cell 32's range, L points
    mu_h, sg_h = make_inputs(rows, 1, seed=1000)
    mu, sg = torch.from_numpy(mu_h.reshape(rows)).to(dev), torch.from_numpy(sg_h.reshape(rows)).to(dev)
    pts_h, lens_h = Emb.make_code_book(Emb.empirical_std(mu), N_BITS)                  # ipynb:373-390
    codebook = torch.from_numpy(pts_h).to(dev)
    idx = torch.empty((L, rows), dtype=torch.uint16, device=dev)
    counts = torch.zeros((L, 1, T), dtype=torch.int64, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        ops.quantize_notebook(mu, sg, codebook, BETAS, N=N_BITS, want_values=False, out_idx=idx)
        if i is not None:
            ev[i][1].record()
        counts.zero_()
        ops.histogram(idx, 1, N=N_BITS, out=counts)

    for _ in range(warmup):
        step()
    clock_ramp(torch, step)                                    # untimed, see clock_ramp
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k1_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    alg = rows * (8 + 2 * L)
    out = {"metric": "quantized latents/sec (32-beta sweep, notebook arithmetic)", "value": rows * L * steps / dt, "unit": "latents/s",
           "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"{name}: {desc}; {L}-point beta sweep exp(linspace(log 0.01, log 1e5, {L})), notebook arithmetic "
                                  f"(K1nt, f64 squared error, ipynb:429-443) + K2 histogram (empirical_entropy, ipynb:452-455); "
                                  f"one solve per (element, beta) per step", "elements_per_gpu": rows, "lambdas": L},
           "roofline": {"bound": "hbm", "limited_by": "valu issue", "kernel": "k_quant_notebook_hull", "achieved": alg / (k1_ms * 1e-3) / 1e9,
                        "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg / (k1_ms * 1e-3) / HBM_PEAK, "traffic": None,
                        "algorithmic_bytes_per_launch": alg, "avg_launch_ms": k1_ms}}
    # the notebook's own brute force over all 2047 code points (C oracle, OpenMP), a bounded sample
    from oracle import c_oracle as CO, vbq_oracle as O
    th = CO.max_threads()
    n = int(min(rows, 400_000 if cpu else 50_000))
    r2s = O.level_major_to_rank(N_BITS)
    t0 = time.perf_counter()
    slots = [CO.compress_coordinates(mu_h[:n, 0], sg_h[:n, 0], b, pts_h, lens_h, threads=th)[1] for b in BETAS]
    dtc = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": n * L / dtc, "unit": "latents/s", "cores": th, "kind": "port",
                           "sample": f"first {n} of {rows} elements x {L} betas, C oracle of compress_coordinates "
                                     f"(2047-point f64 brute force, OpenMP {th} threads), {dtc:.1f} s"}
    got = idx[:, :n].cpu().numpy().astype(np.int64)
    out["parity_vs_oracle_on_sample"] = bool(all(np.array_equal(got[i], r2s[slots[i]]) for i in range(L)))
    return out


if __name__ == "__main__":
    main()

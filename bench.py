#!/usr/bin/env python3
"""Headline benchmark of the VBQ hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one whole entropy-model build of the reference (quantizer.py:82-150) over one batch, as
vbq_amd.pipeline.EntropyModelBuild runs it: [layout change] -> pass 1 (K1t: 32-lambda solve with raw lengths +
bit-length histogram, from 10 thresholds per element instead of a loop over lambda) -> length table -> pass 2 (K1: 32-lambda solve with corrected lengths -> rank indices;
K2: their per-(lambda, channel) histogram) -> code-length models,
and, with N > 1, the RCCL all-reduces of the two histograms.  Every (element, lambda) is therefore solved TWICE per
step; `value` = solves ("quantized latents") per second over all ranks.  Inputs are resident in HBM before the
timed region.  Workload at N = 1 (BASELINE.json configs[1]): the latent tensor of the Kodak-24 set from the
paper's model (--num_filters 256: 24 x 32 x 48 positions x 256 channels = 36864 x 256), 32-point lambda sweep
2**linspace(-8, 7.5, 32), per-channel code books of 2047 points.  Data are synthetic stand-ins of that shape (no
checkpoint / images ship with the reference).  Each rank owns its own batch (weak scaling); the only collectives
are the histogram all-reduces.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (pass 2's K1, k_quant_fast) with its
ALGORITHMIC bytes -- 8 B read per element + 2 B written per (element, lambda) -- against 8 TB/s, from HIP events
around its launches inside the timed steps.  `cpu_baseline` = the C oracle (oracle/vbq_oracle.c, OpenMP) on this
host's cores on a bounded sample of the same workload.  `workloads` repeats the measurement, with parity checks
against the oracle, for the other single-GPU configurations of BASELINE.json.
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
HBM_PEAK = 8.0e12

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
OTHER_WORKLOADS = ["embeddings_1e7", "embeddings_4e5x300", "synthetic_1e8", "shard_1.25e8"]


def make_inputs(rows, C, seed):
    """Synthetic (mu, sigma, level-major tables) with the statistics of SURVEY 8(d)."""
    from scipy.stats import norm
    rng = np.random.default_rng(seed)
    s_c = np.exp(rng.uniform(np.log(0.3), np.log(3.0), C)) if C > 1 else np.array([1.2329])
    m_c = np.zeros(C) if C > 1 else np.array([-0.0799])
    mu = rng.standard_normal((rows, C), dtype=np.float32)
    mu *= s_c.astype(np.float32)
    mu += m_c.astype(np.float32)
    sigma = rng.standard_normal((rows, C), dtype=np.float32)
    sigma *= np.float32(0.7)
    sigma -= np.float32(2.0)
    np.exp(sigma, out=sigma)
    np.clip(sigma, 1e-4, 10, out=sigma)
    scale = np.sqrt(np.mean(mu.astype(np.float64) ** 2, axis=0)) if rows * C <= 50_000_000 else \
        np.sqrt(np.array([np.add.reduce(mu[:, c].astype(np.float64) ** 2) / rows for c in range(C)]))   # empirical prior (ipynb:374)
    xi = np.concatenate([(np.arange(2 ** n) + 0.5) / 2 ** n for n in range(N_BITS + 1)])
    tables = norm.ppf(xi[None, :], scale=scale[:, None]).astype(np.float32)  # [C, T] level-major
    return mu, sigma, tables


def cpu_baseline(mu, sigma, tables, level_len, target_s=12.0):
    from oracle import c_oracle as CO
    threads = CO.max_threads()
    rows, C = mu.shape
    L = len(LAMBDAS)
    probe = max(1, min(rows, 200_000 // C))
    t0 = time.perf_counter()
    CO.quantize(mu[:probe], sigma[:probe], tables, LAMBDAS, N=N_BITS, level_len=level_len, threads=threads)
    dt = time.perf_counter() - t0
    rate = probe * C * L / dt
    n = int(max(probe, min(rows, target_s * rate / (C * L))))
    reps = 0
    t0 = time.perf_counter()
    while reps == 0 or time.perf_counter() - t0 < target_s:      # the probe includes thread start-up: bound by time
        idx = CO.quantize(mu[:n], sigma[:n], tables, LAMBDAS, N=N_BITS, level_len=level_len, threads=threads)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": reps * n * C * L / dt, "unit": "latents/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x (first {n} of {rows} rows x {C} channels x {L} lambdas, corrected lengths), C oracle "
                      f"(oracle/vbq_oracle.c, OpenMP {threads} threads), solve only, {dt:.1f} s"}, idx, n


def cpu_baseline_numpy(mu, sigma, tables, level_len, idx_check, target_s=5.0, max_rows=2048):
    """The reference's own formulation in NumPy (per-level searchsorted on padded grids, 21 x B x C candidate
    tensors, per-lambda argmax: oracle/vbq_oracle.py restating quantizer.py:65-80,156-188 and utils.py:363-423;
    golden vectors G5/G6/G8 tie it to the reference's code), single process, NumPy's default threading."""
    from oracle import vbq_oracle as O
    rows, C = mu.shape
    L = len(LAMBDAS)
    orc = O.ChannelwiseOracle(C, N_BITS)
    orc.build_code_points(lambda xi: tables.T)
    if level_len is not None:
        lv = np.arange(N_BITS + 1, dtype=np.float32)
        orc.raw_models = {lam: level_len[i] - lv[None, :] for i, lam in enumerate(LAMBDAS)}
    n = max(1, min(rows, 64))
    t0 = time.perf_counter()
    orc.compress_batch(mu[:n], sigma[:n], LAMBDAS)
    rate = n * C * L / (time.perf_counter() - t0)
    n = int(max(n, min(rows, max_rows, target_s * rate / (C * L))))      # the 21 x B x C x L length stack bounds B
    t0 = time.perf_counter()
    Z, _ = orc.compress_batch(mu[:n], sigma[:n], LAMBDAS)
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
                    bl.get("workload", "").startswith(workload + ":") and "hbm_bytes_per_launch" in pj["k_quant_fast"]:
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


def run_workload(name, args, torch, dist, dev, rank, world, steps, warmup, detailed):
    """Time `steps` alternations of one workload; returns the result dict (rank 0) or None."""
    from vbq_amd import ops
    from vbq_amd.pipeline import EntropyModelBuild
    rows, C, desc = WORKLOADS[name]
    L = len(LAMBDAS)
    E = rows * C
    mu_h, sg_h, tab_h = make_inputs(rows, C, seed=1000 + rank)
    tab = torch.from_numpy(tab_h).to(dev)
    # Channel-last [rows, C] is how the latents arrive (quantizer.py:90-91).  The kernels work on channel-major
    # planes [C, rows]; the layout change is part of every timed step.
    if C > 1:
        mu_in, sg_in = torch.from_numpy(mu_h).to(dev), torch.from_numpy(sg_h).to(dev)
        mu = torch.empty((C, rows), dtype=torch.float32, device=dev)
        sg = torch.empty((C, rows), dtype=torch.float32, device=dev)
    else:
        mu_in = sg_in = None
        mu, sg = torch.from_numpy(mu_h.reshape(1, rows)).to(dev), torch.from_numpy(sg_h.reshape(1, rows)).to(dev)
    build = EntropyModelBuild(rows, C, LAMBDAS, tab, N=N_BITS, add_n_smoothing=1, global_rows=rows * world,
                              distributed=world > 1, level_group=args.level_group, n_chunks=args.chunks)
    if os.environ.get("VBQ_K1_WG_PER_CU"):
        build.k1_workgroups_per_cu = int(os.environ["VBQ_K1_WG_PER_CU"])
    timers = Timers(torch)
    build.timers = timers

    def step():
        if mu_in is not None:
            timers("layout", 0)
            ops.transpose(mu_in, out=mu)
            ops.transpose(sg_in, out=sg)
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
    torch.cuda.synchronize()
    timers.enabled = True
    dt = timed(steps)
    timers.enabled = False
    for r in build.reducers:
        if r is not None:
            r.check()                  # the packed counters' overflow guard (one 8-byte read, outside the timed loop)
    eager_ms = dt / steps * 1e3
    launch = "eager launches, two streams" if len(build.chunks) > 1 else "eager launches"
    # The step only enqueues stream-ordered work (no allocation, no host synchronisation when the length table is
    # tabulated), so the whole alternation replays from one captured HIP graph: ~20 launches become one.  Per-kernel
    # times above come from the eager steps (events cannot sit inside a graph); the step time from the replays.
    if args.graph and world == 1 and build.lut1 is not None:
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
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                graph.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            launch = f"one HIP graph replay per step (eager launches: {eager_ms:.3f} ms per step)"
        except Exception as e:                     # the eager measurement above stands
            torch.cuda.synchronize()
            launch += f" (HIP graph capture failed: {type(e).__name__})"
    k1_ms, k1_n = timers.total_ms("k1") / steps, timers.launches("k1") // steps
    k1h_ms = timers.total_ms("k1h") / steps
    k2_ms = timers.total_ms("k2") / steps
    alg_bytes = E * (8 + 2 * L)                              # one K1 pass: 8 B in per element, 2 B out per solve
    traffic, traffic_src, valu = committed_counters(name, rows, C, L)
    res = {
        "ms_per_step": dt / steps * 1e3,
        "value": world * 2 * E * L * steps / dt,
        "solves_per_step": 2 * E * L,
        # the same throughput counting every (element, lambda) pair once per build instead of once per pass
        "pairs_per_s": world * E * L * steps / dt,
        "roofline": {"bound": "hbm", "kernel": "k_quant_fast (pass 2: corrected lengths -> rank indices)",
                     "achieved": alg_bytes / (k1_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": alg_bytes / (k1_ms * 1e-3) / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_pass": alg_bytes, "launches_per_pass": k1_n,
                     "algorithmic_bytes_per_launch": alg_bytes / max(k1_n, 1), "avg_launch_ms": k1_ms / max(k1_n, 1),
                     "pass_ms": k1_ms, "latents_per_s_kernel_only": E * L / (k1_ms * 1e-3),
                     # what actually bounds the kernel: share of the SIMDs' issue cycles spent on its VALU instructions
                     "valu_issue_frac": (valu or {}).get("valu_issue_frac"), "valu": valu,
                     # the north star's "HBM-read roofline" prices the UNFUSED per-lambda call, 8 B read per latent,
                     # i.e. a ceiling of 1.0e12 latents/s at 8 TB/s (target 60 % = 6.0e11).  The fused kernels read every element
                     # once per sweep, so this is a throughput ratio against that ceiling, not a bandwidth fraction.
                     "per_lambda_read_roofline": {"bytes_read_per_latent": 8, "ceiling_latents_per_s": HBM_PEAK / 8.0,
                                                  "kernel_frac": E * L / (k1_ms * 1e-3) / (HBM_PEAK / 8.0),
                                                  "step_frac_per_gpu": 2 * E * L * steps / dt / (HBM_PEAK / 8.0)},
                     "note": "K2 of the previous row chunk runs concurrently on a second stream" if k1_n > 1 else None},
        "stages_ms": {"layout_change": timers.total_ms("layout") / steps if C > 1 else None,
                      "pass1_k1t_solve_and_level_histogram": k1h_ms, "pass2_k1_solve": k1_ms,
                      "pass2_k2_histogram (overlapped with k1)" if k1_n > 1 else "pass2_k2_histogram": k2_ms},
        "roofline_k1h": {"bound": "hbm", "kernel": "k_level_counts_hull (pass 1, K1t: thresholds instead of a lambda loop; no per-element output)",
                         "achieved": 8.0 * E / (k1h_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": 8.0 * E / (k1h_ms * 1e-3) / HBM_PEAK, "algorithmic_bytes_per_launch": 8 * E,
                         "avg_launch_ms": k1h_ms, "latents_per_s_kernel_only": E * L / (k1h_ms * 1e-3),
                         "note": "reads 8 B per element and writes nothing per element: instruction-bound by construction"},
        "roofline_k2_histogram": {"bound": "hbm", "kernel": "k_hist_flat", "achieved": 2.0 * L * E / (k2_ms * 1e-3) / 1e9,
                                  "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": 2.0 * L * E / (k2_ms * 1e-3) / HBM_PEAK,
                                  "algorithmic_bytes_per_pass": 2 * L * E, "pass_ms": k2_ms},
        "config": {"workload": f"{name}: {desc}; {L}-point lambda sweep 2**linspace(-8,7.5,32); N={N_BITS} (2047 code points/"
                               f"channel); step = two-pass entropy-model build (quantizer.py:82-150): "
                               f"{'layout change + ' if C > 1 else ''}pass 1 (solve, raw lengths, bit-length histogram) + length table + "
                               f"pass 2 (solve, corrected lengths, rank indices + rank histogram"
                               f"{', K2 overlapped in ' + str(len(build.chunks)) + ' row chunks' if len(build.chunks) > 1 else ''}) + models"
                               f"{' + RCCL all-reduce of both histograms' if world > 1 else ''}; 2 solves per (element, lambda) per step",
                   "elements_per_gpu": E, "lambdas": L, "parallelism": f"element-sharded x{world}",
                   "launch": launch,
                   "length_table": "device (tabulated -log2)" if build.lut1 is not None else "host round trip ([L, C, N+1] only)",
                   "models": "device (tabulated -log2)" if build.lut2 is not None else "not in the step (counts stay on the device)"},
    }
    if world > 1:
        # what the collectives cost: the same steps without them, and the rank-histogram all-reduce in isolation
        build.collectives = False
        for _ in range(2):
            step()
        dt0 = timed(max(3, steps // 3)) / max(3, steps // 3)
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
                            "overlap": "asynchronous, two buffers: step i's rank histogram is reduced while step i+1 computes; "
                                       "the bit-length histogram has its own communicator"}

    if not detailed or rank != 0:
        del build
        torch.cuda.empty_cache()
        return res if rank == 0 else None

    # ------------------------------------------------------------ parity of the timed configuration (rank 0)
    from oracle import c_oracle as CO, vbq_oracle as O
    th = CO.max_threads()
    torch.cuda.synchronize()
    ll_h = build.level_len.cpu().numpy()
    parity = {}
    # (1) pass-2 indices against the C oracle on windows at the start, across the middle and at the end
    win = min(rows, max(1, 200_000 // C))
    starts = sorted({0, max(0, rows // 2 - win // 2), rows - win})
    ok = True
    for s in starts:
        want = CO.quantize(mu_h[s:s + win], sg_h[s:s + win], tab_h, LAMBDAS, N=N_BITS, level_len=ll_h, threads=th)   # [L, win, C]
        got = build.idx[:, :, s:s + win].permute(0, 2, 1).cpu().numpy()
        ok = ok and bool(np.array_equal(got, want))
    parity["pass2_indices_equal_oracle_on_windows"] = ok
    parity["windows"] = f"{len(starts)} x {win} rows x {C} channels x {L} lambdas"
    # (2) both histograms against a second route on the device, full size: K1 indices (raw lengths) -> K2 -> level sums
    #     must equal pass 1's K1t counts; K2 of the stored pass-2 indices in one launch must equal the chunked counts
    from vbq_amd import entropy
    lc = build.level_counts.clone()
    cnt = build.counts.clone()
    if world > 1:
        parity["histograms"] = "skipped at N > 1 (global sums)"
    else:
        idx_raw = ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb")
        via = entropy.level_counts_from_counts(ops.histogram(idx_raw, C, N=N_BITS, layout="cb"), N_BITS)
        parity["pass1_level_counts_equal_k1_k2_route"] = bool(torch.equal(via, lc))
        del idx_raw
        one = ops.histogram(build.idx, C, N=N_BITS, layout="cb")
        parity["pass2_rank_counts_equal_single_launch"] = bool(torch.equal(one, cnt.to(torch.int64)))
        parity["counts_total"] = bool(int(cnt.sum().item()) == E * L and int(lc.sum().item()) == E * L)
        # (3) the length table and the models against the reference's NumPy float32 arithmetic on the same counts
        parity["length_table_equals_numpy"] = bool(np.array_equal(
            ll_h, (np.arange(N_BITS + 1, dtype=np.float32) + entropy.neg_log2_freq(lc, 1)).astype(np.float32)))
        if build.models is not None:
            parity["models_equal_numpy"] = bool(np.array_equal(build.models.cpu().numpy(), entropy.neg_log2_freq(cnt, 1)))
        # (4) level counts of pass 1 against the oracle on a sample (exact on the first window's rows)
        w0 = O.levels_of_sorted_ranks(N_BITS)[CO.quantize(mu_h[:win], sg_h[:win], tab_h, LAMBDAS, N=N_BITS, threads=th)]
        want_lc = np.stack([[np.bincount(w0[l, :, c], minlength=N_BITS + 1) for c in range(C)] for l in range(L)])
        got_lc = ops.level_counts(mu[:, :win].contiguous(), sg[:, :win].contiguous(), tab, LAMBDAS, N=N_BITS, layout="cb")
        parity["pass1_level_counts_equal_oracle_on_sample"] = bool(np.array_equal(got_lc.cpu().numpy(), want_lc))
    res["parity"] = parity
    res["parity_ok"] = all(v for v in parity.values() if isinstance(v, bool))
    res["_host"] = (mu_h, sg_h, tab_h, ll_h, build.idx)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="kodak24_c256", choices=sorted(WORKLOADS))
    ap.add_argument("--chunks", type=int, default=None, help="row chunks of pass 2 (K2 of chunk j overlaps K1 of chunk j+1 on a second "
                                                             "stream); default 1: the overlap measured slower, profiles/r2_overlap_sweep.txt")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="time eager launches instead of replaying the step from a captured HIP graph (one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the `workloads` section (the other BASELINE configs)")
    ap.add_argument("--notebook", action="store_true",
                    help="C = 1 workloads in the word-embedding notebook's arithmetic (K1n: f64 squared error, penalty "
                         "fl32(2 beta sigma^2) * length, ipynb:429-443): one K1n + K2 pass per step")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs a torch.distributed.run launch with {args.gpus} ranks")
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

    if args.notebook:
        out = run_notebook(args, torch, dev, cpu=not args.no_cpu_baseline)
        print(json.dumps(out))
        return

    res = run_workload(args.workload, args, torch, dist, dev, rank, world, args.steps, args.warmup, detailed=(world == 1))
    out = None
    if rank == 0:
        host = res.pop("_host", None)
        out = {
            "metric": "quantized latents/sec (32-lambda sweep)",
            "value": res["value"],
            "unit": "latents/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": res["config"],
            "roofline": res["roofline"],
            "solves_per_step": res["solves_per_step"],
            "pairs_per_s": res["pairs_per_s"],
            "stages_ms": res["stages_ms"],
            "roofline_k1h": res["roofline_k1h"],
            "roofline_k2_histogram": res["roofline_k2_histogram"],
        }
        if "allreduce" in res:
            out["allreduce"] = res["allreduce"]
        if "parity" in res:
            out["parity"] = res["parity"]
            out["parity_vs_oracle_on_sample"] = res["parity_ok"]
        if world == 1 and not args.no_cpu_baseline and host is not None:
            mu_h, sg_h, tab_h, ll_h, idx = host
            C = mu_h.shape[1]
            cb, idx_cpu, n = cpu_baseline(mu_h, sg_h, tab_h, ll_h)
            out["cpu_baseline"] = cb
            # the sample doubles as a second in-run parity check of the timed configuration (pass 2)
            got = idx[:, :, :n].permute(0, 2, 1).cpu().numpy()
            out["parity_vs_oracle_on_sample"] = bool(out.get("parity_vs_oracle_on_sample", True) and np.array_equal(got, idx_cpu))
            # the same formulation in NumPy, as the reference runs it (reported next to the C port, never the target)
            out["cpu_baseline_numpy"] = cpu_baseline_numpy(mu_h, sg_h, tab_h, ll_h, idx_cpu if C > 1 else None)
        else:
            out["cpu_baseline"] = None
        del host
    del res
    torch.cuda.empty_cache()

    # the other BASELINE configurations, same step, same checks, fewer repetitions (every rank takes part when N > 1)
    if not args.no_other_workloads:
        names = [w for w in OTHER_WORKLOADS if w != args.workload] if world == 1 else \
                [w for w in ["shard_1.25e8"] if w != args.workload]
        others = {}
        for w in names:
            r = run_workload(w, args, torch, dist, dev, rank, world, steps=max(3, min(args.steps, 5)), warmup=2, detailed=(world == 1))
            if rank == 0:
                r.pop("_host", None)
                others[w] = {"ms_per_step": r["ms_per_step"], "value": r["value"], "unit": "latents/s",
                             "roofline": {k: r["roofline"][k] for k in ("kernel", "achieved", "peak", "unit", "frac", "pass_ms", "launches_per_pass")},
                             "stages_ms": r["stages_ms"], "k1h_latents_per_s": r["roofline_k1h"]["latents_per_s_kernel_only"],
                             "k2_frac": r["roofline_k2_histogram"]["frac"], "elements_per_gpu": r["config"]["elements_per_gpu"],
                             "workload": r["config"]["workload"], "length_table": r["config"]["length_table"],
                             "parity": r.get("parity"), "parity_ok": r.get("parity_ok"), "allreduce": r.get("allreduce")}
            torch.cuda.empty_cache()
        if rank == 0 and world == 1:
            nb = run_notebook(args, torch, dev, workload="embeddings_1e7", steps=5, warmup=2, cpu=False)
            others["embeddings_1e7_notebook"] = {k: nb[k] for k in ("ms_per_step", "value", "unit", "roofline", "parity_vs_oracle_on_sample")}
            others["embeddings_1e7_notebook"]["workload"] = nb["config"]["workload"]
            others["kodak24_c256_quantize_sweep_raw"] = run_raw_sweep(torch, dev)
        if rank == 0:
            out["workloads"] = others
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def run_raw_sweep(torch, dev, steps=10, warmup=3):
    """The literal north-star call, quantize(mu, sigma, lmbda) for the 32-point sweep with the reference's raw code lengths
    (quantizer.py:167-169): ONE pass, indices out, no histogram -- K1e (thresholds + a walk down the staircase)."""
    from vbq_amd import ops
    from oracle import c_oracle as CO
    rows, C, desc = WORKLOADS["kodak24_c256"]
    mu_h, sg_h, tab_h = make_inputs(rows, C, seed=1000)
    mu = torch.from_numpy(np.ascontiguousarray(mu_h.T)).to(dev)
    sg = torch.from_numpy(np.ascontiguousarray(sg_h.T)).to(dev)
    tab = torch.from_numpy(tab_h).to(dev)
    L = len(LAMBDAS)
    idx = torch.empty((L, C, rows), dtype=torch.uint16, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for _ in range(warmup):
        ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb", out_idx=idx)
    for a, b in ev:
        a.record()
        ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb", out_idx=idx)
        b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    alg = rows * C * (8 + 2 * L)
    n = 2048                                                   # parity: the first rows of every channel against the C oracle
    want = CO.quantize(mu_h[:n], sg_h[:n], tab_h, LAMBDAS, N=N_BITS, threads=CO.max_threads())       # [L, n, C]
    ok = bool(np.array_equal(idx[:, :, :n].cpu().numpy().transpose(0, 2, 1), want))
    return {"ms_per_step": ms, "value": rows * C * L / (ms * 1e-3), "unit": "latents/s",
            "roofline": {"bound": "hbm", "kernel": "k_quant_hull_idx", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": alg / (ms * 1e-3) / HBM_PEAK, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms},
            "parity_vs_oracle_on_sample": ok,
            "workload": f"kodak24_c256: {desc}; one pass of quantize(mu, sigma, lmbda) over the {L}-point sweep, raw code lengths, "
                        f"channel-major planes in, rank indices out"}


def run_notebook(args, torch, dev, workload=None, steps=None, warmup=None, cpu=True):
    """The C = 1 workloads in the notebook's own arithmetic: one K1n launch + one K2 launch per step."""
    from vbq_amd import ops, embeddings as Emb
    name = workload or (args.workload if WORKLOADS[args.workload][1] == 1 else "embeddings_1e7")
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    rows, C, desc = WORKLOADS[name]
    L = len(LAMBDAS)
    BETAS = [float(b) for b in np.exp(np.linspace(np.log(0.01), np.log(1e5), L))]      # ipynb cell 32's range, L points
    mu_h, sg_h, _ = make_inputs(rows, 1, seed=1000)
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
           "roofline": {"bound": "hbm", "kernel": "k_quant_notebook_hull", "achieved": alg / (k1_ms * 1e-3) / 1e9,
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

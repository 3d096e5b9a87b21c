#!/usr/bin/env python3
"""Headline benchmark of the VBQ hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one entropy-model pass of the reference (quantizer.py:119-146) over one batch:
the fused 32-lambda R-D solve (K1), the per-(lambda, channel) histogram of the indices (K2)
and, with N > 1, the RCCL all-reduce of that histogram.  Inputs are resident in HBM before
the timed region.  Workload (BASELINE.json configs[1]): the latent tensor of the Kodak-24
set from the paper's model (--num_filters 256: 24 x 32 x 48 positions x 256 channels =
36864 x 256), 32-point lambda sweep 2**linspace(-8, 7.5, 32), per-channel code books of
2047 points, corrected code lengths (the production compress path, quantizer.py:171-180).
Data are synthetic stand-ins of that shape (no checkpoint / images ship with the reference).
Each rank owns its own batch (weak scaling); the only collective is the histogram all-reduce.

Prints ONE JSON line (rank 0).  `value` = quantized latents (element x lambda solves) per
second over all ranks.  `roofline` prices the dominant kernel (K1) with its ALGORITHMIC
bytes -- 8 B read per element + 2 B written per (element, lambda) -- against 8 TB/s.
`cpu_baseline` = the C oracle (oracle/vbq_oracle.c, OpenMP) on this host's cores on a
bounded sample of the same workload.
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
    "kodak24_c256": (36864, 256, "Kodak-24 latents of bls2017 (num_filters=256): [36864 x 256] f32"),
    "kodak24_c32": (36864, 32, "Kodak-24 latents of bls2017 (default num_filters=32): [36864 x 32] f32"),
    "embeddings_1e7": (10_000_000, 1, "word embeddings 100000 x 100, one Gaussian code book"),
    "embeddings_4e5x300": (120_000_000, 1, "word embeddings 4e5 x 300 (BASELINE.json configs[2]), one Gaussian code book"),
    "synthetic_1e8": (100_000_000, 1, "synthetic 1e8-element tensor, one code book (configs[3])"),
    "shard_1.25e8": (125_000_000, 1, "one rank's 1.25e8-element shard of the 1e9-element tensor (configs[4])"),
}


def make_inputs(rows, C, seed):
    """Synthetic (mu, sigma, level-major tables) with the statistics of SURVEY 8(d)."""
    from scipy.stats import norm
    rng = np.random.default_rng(seed)
    s_c = np.exp(rng.uniform(np.log(0.3), np.log(3.0), C)) if C > 1 else np.array([1.2329])
    m_c = np.zeros(C) if C > 1 else np.array([-0.0799])
    mu = (m_c + s_c * rng.standard_normal((rows, C), dtype=np.float32)).astype(np.float32)
    sigma = np.clip(np.exp(-2.0 + 0.7 * rng.standard_normal((rows, C), dtype=np.float32)), 1e-4, 10).astype(np.float32)
    scale = np.sqrt(np.mean(mu.astype(np.float64) ** 2, axis=0))           # empirical prior (ipynb:374)
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
    reps = int(max(1, round(target_s * rate / (n * C * L))))
    t0 = time.perf_counter()
    for _ in range(reps):
        idx = CO.quantize(mu[:n], sigma[:n], tables, LAMBDAS, N=N_BITS, level_len=level_len, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": reps * n * C * L / dt, "unit": "latents/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x (first {n} of {rows} rows x {C} channels x {L} lambdas), C oracle "
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="kodak24_c256", choices=sorted(WORKLOADS))
    ap.add_argument("--stage", default="full", choices=["full", "quantize"])
    ap.add_argument("--raw-lengths", action="store_true", help="pass-1 lengths (n) instead of corrected lengths")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--notebook", action="store_true",
                    help="C = 1 workloads in the word-embedding notebook's arithmetic (K1n: f64 squared error, penalty "
                         "fl32(2 beta sigma^2) * length, ipynb:429-443) instead of the image pipeline's f32 score")
    ap.add_argument("--direct-bc", action="store_true",
                    help="C > 1: let K1 read the channel-last inputs itself (VBQ_LAYOUT_BC_TO_CB) instead of transposing them "
                         "into planes first; measured 1-2 %% slower end to end (K1 +36 us against 31 us of transposes)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a captured HIP graph (one GPU; per-kernel times come from an eager pass)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from vbq_amd import ops

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

    rows, C, desc = WORKLOADS[args.workload]
    L = len(LAMBDAS)
    if args.notebook and C != 1:
        raise SystemExit("--notebook applies to the one-code-book (C = 1) workloads")
    BETAS = [float(b) for b in np.exp(np.linspace(np.log(0.01), np.log(1e5), L))]      # ipynb cell 32's range, L points
    mu_h, sg_h, tab_h = make_inputs(rows, C, seed=1000 + rank)
    mu, sg, tab = torch.from_numpy(mu_h).to(dev), torch.from_numpy(sg_h).to(dev), torch.from_numpy(tab_h).to(dev)
    E = rows * C
    # Channel-last [rows, C] is how the latents arrive (quantizer.py:90-91).  The kernels work on
    # channel-major planes [C, rows]; the layout change is part of every timed step.
    if C > 1:
        mu_in, sg_in = mu.reshape(rows, C), sg.reshape(rows, C)
        mu = torch.empty((C, rows), dtype=torch.float32, device=dev)
        sg = torch.empty((C, rows), dtype=torch.float32, device=dev)
        ops.transpose(mu_in, out=mu)
        ops.transpose(sg_in, out=sg)
        shape, layout = (C, rows), "cb"
    else:
        mu_in = sg_in = None
        mu, sg = mu.reshape(rows), sg.reshape(rows)
        shape, layout = (rows,), "bc"
    idx = torch.empty((L,) + shape, dtype=torch.uint16, device=dev)
    # int32 counters halve the all-reduce payload; exact while the global rows per channel < 2^31
    cdtype = torch.int32 if rows * world < 2 ** 31 else torch.int64
    counts2 = [torch.zeros((L, C, T), dtype=cdtype, device=dev) for _ in range(2)]
    works = [None, None]
    reducers = None
    if world > 1:
        from vbq_amd.dist import CountsAllReduce
        reducers = [CountsAllReduce(L * C * T, dev, max_global_count=rows * world) if cdtype == torch.int32 else None for _ in range(2)]
    ws = torch.empty(ops._lib.lib().vbq_quantize_workspace_bytes(C, L, N_BITS), dtype=torch.uint8, device=dev)

    codebook = None
    if args.notebook:
        from vbq_amd import embeddings as Emb
        pts_h, lens_h = Emb.make_code_book(Emb.empirical_std(mu), N_BITS)              # ipynb:373-390
        codebook = torch.from_numpy(pts_h).to(dev)
    # setup (untimed): pass 1 with raw lengths -> bit-length histogram -> corrected lengths (quantizer.py:96-112)
    level_len = None
    if not args.raw_lengths and not args.notebook:
        from vbq_amd.entropy import level_lengths_from_counts
        ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, layout=layout, out_idx=idx, workspace=ws)
        c1 = ops.histogram(idx, C, N=N_BITS, layout=layout)
        if world > 1:
            dist.all_reduce(c1)
        level_len = level_lengths_from_counts(c1, N_BITS, add_n_smoothing=1)          # f32 [L, C, N+1] on device

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    evh = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None, slot=0):
        counts = counts2[slot]
        if works[slot] is not None:          # the all-reduce that last used this buffer must be done
            works[slot].wait()
            works[slot] = None
        if mu_in is not None and not args.direct_bc:
            ops.transpose(mu_in, out=mu)
            ops.transpose(sg_in, out=sg)
        if i is not None:
            ev[i][0].record()
        if mu_in is not None and args.direct_bc:
            # K1 reads the latents as they arrive (channel-last) and writes channel-major planes
            ops.quantize(mu_in, sg_in, tab, LAMBDAS, N=N_BITS, level_len=level_len, layout="bc->cb", out_idx=idx, workspace=ws)
        elif args.notebook:
            ops.quantize_notebook(mu, sg, codebook, BETAS, N=N_BITS, want_values=False, out_idx=idx)
        else:
            ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, level_len=level_len, layout=layout, out_idx=idx, workspace=ws)
        if i is not None:
            ev[i][1].record()
        if args.stage == "full":
            counts.zero_()
            if i is not None:
                evh[i][0].record()
            ops.histogram(idx, C, N=N_BITS, layout=layout, out=counts)
            if i is not None:
                evh[i][1].record()
            if world > 1:
                # asynchronous: the collective of step i overlaps the kernels of step i+1 (two buffers)
                # (three 21-bit counters per int64 word on the wire while the global counts allow it)
                works[slot] = reducers[slot].start(counts) if reducers[slot] is not None else dist.all_reduce(counts, async_op=True)

    def drain():
        for b in range(2):
            if works[b] is not None:
                works[b].wait()
                works[b] = None

    for w in range(args.warmup):
        step(None, w & 1)
    drain()
    torch.cuda.synchronize()
    graph = None
    if args.graph:
        if world > 1:
            raise SystemExit("--graph is a single-GPU option")
        for i in range(args.steps):                   # per-kernel times (HIP events cannot sit inside the graph)
            step(i, 0)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        cs = torch.cuda.Stream()
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            step(None, 0)
        torch.cuda.current_stream().wait_stream(cs)
        with torch.cuda.graph(graph):
            step(None, 0)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            step(i, i & 1)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # HBM traffic of K1 per launch from the committed PMC passes (profiles/*_pmc.json): 2 x FETCH_SIZE +
    # WRITE_SIZE, collected with rocprofv3 --pmc in separate runs of this same workload.
    traffic, traffic_src, valu_note = None, None, None
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
            pj = json.load(open(f))
            bl = pj.get("bench_line", {}).get("config", {})
            if bl.get("elements_per_gpu") == rows * C and bl.get("lambdas") == L and "k_quant_fast" in pj and \
                    bl.get("workload", "").startswith(args.workload + ":") and \
                    "hbm_bytes_per_launch" in pj["k_quant_fast"]:
                traffic, traffic_src = pj["k_quant_fast"]["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
                sqc = pj["k_quant_fast"].get("sq", {})
                if "SQ_INSTS_VALU" in sqc:          # why the HBM fraction is what it is: the kernel is VALU-issue-bound
                    valu_note = {"valu_wave_instructions_per_launch": sqc["SQ_INSTS_VALU"],
                                 "valu_instructions_per_latent": sqc["SQ_INSTS_VALU"] * 64.0 / (rows * C * L),
                                 "clock_GHz_under_load": sqc.get("clock_GHz"), "source": traffic_src}
                break
    except Exception:
        pass
    k1_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    k2_ms = float(np.mean([a.elapsed_time(b) for a, b in evh])) if args.stage == "full" else None
    alg_bytes = E * (8 + 2 * L)
    achieved = alg_bytes / (k1_ms * 1e-3)

    out = None
    if rank == 0:
        out = {
            "metric": "quantized latents/sec (32-lambda sweep)",
            "value": world * E * L * args.steps / dt,
            "unit": "latents/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}; "
                                   + (f"{L}-point beta sweep exp(linspace(log 0.01, log 1e5, {L})), notebook arithmetic (K1n, f64 "
                                      f"squared error, ipynb:429-443); " if args.notebook else
                                      f"{L}-point lambda sweep 2**linspace(-8,7.5,32); ")
                                   + f"N={N_BITS} (2047 code points/channel); "
                                   f"{'raw' if args.raw_lengths or args.notebook else 'corrected'} code lengths; stage={args.stage} "
                                   f"({'layout change + ' if C > 1 and not args.direct_bc else ''}K1 solve{' (channel-last in, planes out)' if C > 1 and args.direct_bc else ''}{' + K2 histogram' + (' + RCCL all-reduce' if world > 1 else '') if args.stage == 'full' else ''})",
                       "elements_per_gpu": E, "lambdas": L, "parallelism": f"element-sharded x{world}",
                       "launch": "hip graph replay" if args.graph else "eager launches"},
            "roofline": {"bound": "hbm", "kernel": "k_quant_notebook_fast" if args.notebook else "k_quant_fast",
                         "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": k1_ms,
                         "latents_per_s_kernel_only": E * L / (k1_ms * 1e-3), "valu": valu_note},
            "stages_ms": {"k1_solve": k1_ms, "k2_histogram": k2_ms},
            "roofline_k2_histogram": (None if k2_ms is None else
                                      {"bound": "hbm", "kernel": "k_hist_flat", "achieved": 2.0 * L * E / (k2_ms * 1e-3) / 1e9,
                                       "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": 2.0 * L * E / (k2_ms * 1e-3) / HBM_PEAK,
                                       "algorithmic_bytes_per_launch": 2 * L * E, "avg_launch_ms": k2_ms}),
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.notebook:
        # the notebook's own brute force over all 2047 code points (C oracle, OpenMP), a bounded sample
        from oracle import c_oracle as CO, vbq_oracle as O
        th = CO.max_threads()
        n = int(min(rows, 400_000))
        r2s = O.level_major_to_rank(N_BITS)
        t0 = time.perf_counter()
        slots = [CO.compress_coordinates(mu_h[:n, 0], sg_h[:n, 0], b, pts_h, lens_h, threads=th)[1] for b in BETAS]
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n * L / dtc, "unit": "latents/s", "cores": th, "kind": "port",
                               "sample": f"first {n} of {rows} elements x {L} betas, C oracle of compress_coordinates "
                                         f"(2047-point f64 brute force, OpenMP {th} threads), {dtc:.1f} s"}
        got = idx[:, :n].cpu().numpy().astype(np.int64)
        out["parity_vs_oracle_on_sample"] = bool(all(np.array_equal(got[i], r2s[slots[i]]) for i in range(L)))
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        ll_h = level_len.cpu().numpy() if level_len is not None else None
        cb, idx_cpu, n = cpu_baseline(mu_h, sg_h, tab_h, ll_h)
        out["cpu_baseline"] = cb
        # the sample doubles as an in-run parity check of the timed configuration
        got = (idx[:, :, :n].permute(0, 2, 1) if C > 1 else idx[:, :n]).cpu().numpy().reshape(idx_cpu.shape)
        out["parity_vs_oracle_on_sample"] = bool(np.array_equal(got, idx_cpu))
        # the same formulation in NumPy, as the reference runs it (reported next to the C port, never the target)
        out["cpu_baseline_numpy"] = cpu_baseline_numpy(mu_h, sg_h, tab_h, ll_h, idx_cpu.reshape(L, -1, C) if C > 1 else None)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Sharded entropy-model build on the GPU: two ranks (both on cuda:0, gloo as the transport since
the box has one GPU) each own half of the rows; with process_group set, the histograms are
all-reduced and every rank ends with the models a single process computes from all rows."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 10
LAMBS = [float(v) for v in 2.0 ** np.linspace(-6, 6, 5)]


def _data():
    rng = np.random.default_rng(12)
    C, B = 5, 3001
    scale = np.exp(rng.uniform(np.log(0.3), np.log(3.0), C))
    mu = (scale * rng.normal(0, 1, (B, C))).astype(np.float32)
    sg = np.exp(rng.normal(-2, 0.7, (B, C))).astype(np.float32)
    return scale, mu, sg


def _build(mu, sg, scale, group=None):
    from vbq_amd import ChannelwisePriorCDFQuantizer, priors
    q = ChannelwisePriorCDFQuantizer(len(scale), N)
    q.build_code_points(priors.FactoredGaussianPrior(np.zeros(len(scale)), scale))
    q.process_group = group
    q.build_entropy_models_from_latents(mu, sg, LAMBS, add_n_smoothing=1)
    return q


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from vbq_amd import dist as vd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    scale, mu, sg = _data()
    a, b = vd.shard_rows(mu.shape[0], rank, world)
    q = _build(mu[a:b], sg[a:b], scale, group=dist.group.WORLD)
    out.put((rank, {l: q.raw_code_length_entropy_models[l] for l in LAMBS}, {l: q.entropy_models[l] for l in LAMBS}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_entropy_models_equal_single_process():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    scale, mu, sg = _data()
    ref = _build(mu, sg, scale)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29300 + os.getpid() % 500
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, raw, full in res:
        for l in LAMBS:
            assert np.array_equal(raw[l], ref.raw_code_length_entropy_models[l])
            assert np.array_equal(full[l], ref.entropy_models[l])


def test_pack_counts_round_trip_and_field_sums():
    """vbq_pack_counts_3x21: unpack(pack(c)) == c for ragged sizes, and the SUM of packed words is the field-wise
    sum while every total stays below 2^21 (what the all-reduce relies on)."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    from vbq_amd import _lib, ops
    h = _lib.lib()
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 4, 1000, 3 * 2047 + 1):
        parts = [torch.from_numpy(rng.integers(0, (1 << 21) // 8, n).astype(np.int32)).cuda() for _ in range(8)]
        parts[0][0] = (1 << 21) // 8 - 1
        words = []
        for c in parts:
            w = torch.empty((n + 2) // 3, dtype=torch.int64, device="cuda")
            flag = torch.zeros(1, dtype=torch.uint32, device="cuda")
            _lib.check(h.vbq_pack_counts_3x21(ops._ptr(c), n, ops._ptr(w), 8, ops._ptr(flag), ops._stream(c)), "pack")
            assert int(flag.cpu().item()) == 0
            back = torch.empty_like(c)
            _lib.check(h.vbq_unpack_counts_3x21(ops._ptr(w), n, ops._ptr(back), ops._stream(c)), "unpack")
            assert torch.equal(back, c)
            words.append(w)
        total = torch.stack(words).sum(0)
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        _lib.check(h.vbq_unpack_counts_3x21(ops._ptr(total), n, ops._ptr(out), ops._stream(out)), "unpack")
        assert torch.equal(out, torch.stack(parts).sum(0).to(torch.int32))


def test_pack_counts_overflow_guard():
    """A local count at or above 2^21 / n_ranks (or negative) raises the device flag; below it the flag stays 0."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    from vbq_amd import _lib, ops
    h = _lib.lib()
    for world, bad, pos in ((2, 1 << 20, 5), (8, (1 << 21) // 8, 0), (1, -1, 7), (4, (1 << 21) // 4 - 1, 3)):
        c = torch.full((10,), 5, dtype=torch.int32, device="cuda")
        c[pos] = bad
        w = torch.empty(4, dtype=torch.int64, device="cuda")
        flag = torch.zeros(1, dtype=torch.uint32, device="cuda")
        _lib.check(h.vbq_pack_counts_3x21(ops._ptr(c), 10, ops._ptr(w), world, ops._ptr(flag), ops._stream(c)), "pack")
        expect = 0 if (0 <= bad < (1 << 21) // world) else 1
        assert int(flag.cpu().item()) == expect, (world, bad)


def _reduce_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from vbq_amd.dist import CountsAllReduce
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(100 + rank)
    res = []
    for limit in (1000, 1 << 22):                                   # packed and plain int32 paths
        c = torch.from_numpy(rng.integers(0, 400, (3, 4, 2047)).astype(np.int32)).to(dev)
        red = CountsAllReduce(c.numel(), dev, max_global_count=limit)
        red.start(c).wait(check=True)
        torch.cuda.synchronize()
        res.append((red.packed, c.cpu().numpy()))
    # a count that could carry into the next field: every rank must see the flag (rank 1 alone holds the count)
    c = torch.full((6,), 3, dtype=torch.int32, device=dev)
    if rank == 1:
        c[2] = 1 << 20
    red = CountsAllReduce(c.numel(), dev, max_global_count=1000)
    red.start(c).wait()
    try:
        red.check()
        res.append("no error")
    except Exception as e:
        res.append(type(e).__name__)
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_counts_all_reduce_packed_equals_plain():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29800 + os.getpid() % 150
    procs = [ctx.Process(target=_reduce_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = []
    for i in range(2):
        gens = [np.random.default_rng(100 + r) for r in range(2)]
        arrs = []
        for g in gens:
            a = [g.integers(0, 400, (3, 4, 2047)).astype(np.int32) for _ in range(2)]
            arrs.append(a[i])
        want.append(arrs[0] + arrs[1])
    for r in range(2):
        assert res[r][0][0] is True and res[r][1][0] is False
        assert res[r][2] == "VBQError"
        for i in range(2):
            assert np.array_equal(res[r][i][1], want[i])


def _rccl_single_rank_worker(port, out):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from vbq_amd.dist import CountsAllReduce
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    rng = np.random.default_rng(7)
    c = torch.from_numpy(rng.integers(0, 30000, (4, 3, 2047)).astype(np.int32)).to(dev)
    want = c.clone()
    ok = []
    for limit in (36864, 1 << 30):                                  # packed int64 words, then plain int32
        red = CountsAllReduce(c.numel(), dev, max_global_count=limit)
        red.start(c).wait()
        torch.cuda.synchronize()
        ok.append(bool(torch.equal(c, want)))
    t = torch.ones(5, dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    ok.append(bool(torch.equal(t.cpu(), torch.ones(5, dtype=torch.float64))))
    dist.barrier()
    dist.destroy_process_group()
    out.put(ok)


@pytest.mark.timeout(300)
def test_rccl_backend_single_rank():
    """The collectives of the N > 1 path on the real backend (RCCL, one rank: the box has one GPU): process-group
    creation with device_id, int64 / int32 SUM all-reduce through CountsAllReduce, f64 all-reduce, barrier."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    import torch.distributed as dist
    if not dist.is_nccl_available():
        pytest.skip("torch.distributed was built without the nccl (RCCL) backend")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank_worker, args=(29950 + os.getpid() % 40, out))
    p.start()
    res = out.get(timeout=240)
    p.join(60)
    assert p.exitcode == 0 and res == [True, True, True]


# ---------------------------------------------------------------------------------------------------------------
# The C-ABI's own communicator (vbq_comm_* / vbq_allreduce_hist) and the real two-GPU path.
# Children are fresh interpreters (spawn) that pick their GPU before any HIP call; the parent only COUNTS devices.
def _cabi_comm_worker(rank, world, idfile, out):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import time
    torch.cuda.set_device(rank)
    from vbq_amd.dist import Communicator
    if rank == 0:
        uid = Communicator.unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        for _ in range(600):
            if os.path.exists(idfile):
                break
            time.sleep(0.1)
        uid = open(idfile, "rb").read()
    comm = Communicator(world, rank, uid)
    rng = np.random.default_rng(40 + rank)
    c64 = torch.from_numpy(rng.integers(0, 1 << 40, (3, 2, 2047)).astype(np.int64)).cuda()
    c32 = torch.from_numpy(rng.integers(0, 1 << 20, (3, 2, 11)).astype(np.int32)).cuda()
    comm.all_reduce_(c64)
    comm.all_reduce_(c32)
    torch.cuda.synchronize()
    out.put((rank, c64.cpu().numpy(), c32.cpu().numpy()))
    comm.close()


def _run_cabi_comm(world, tmp_path):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    idfile = str(tmp_path / "rccl_id.bin")
    procs = [ctx.Process(target=_cabi_comm_worker, args=(r, world, idfile, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (a, b)) for r, a, b in (out.get(timeout=240) for _ in range(world)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    gens = [np.random.default_rng(40 + r) for r in range(world)]
    w64 = sum(g.integers(0, 1 << 40, (3, 2, 2047)).astype(np.int64) for g in gens)
    gens = [np.random.default_rng(40 + r) for r in range(world)]
    for g in gens:
        g.integers(0, 1 << 40, (3, 2, 2047))
    w32 = sum(g.integers(0, 1 << 20, (3, 2, 11)).astype(np.int32) for g in gens)
    for r in range(world):
        assert np.array_equal(res[r][0], w64) and np.array_equal(res[r][1], w32)


@pytest.mark.timeout(300)
def test_cabi_communicator_single_rank(tmp_path):
    """vbq_comm_unique_id / vbq_comm_init / vbq_allreduce_hist / vbq_comm_destroy through RCCL (bound with dlopen) on the
    one GPU every box has: the all-reduce over one rank must leave int64 and int32 histograms untouched."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    _run_cabi_comm(1, tmp_path)


def _run_bench_two_ranks(extra, tmp_path, launcher="torchrun"):
    """-> (the compact line the driver parses, the full record from the side file)."""
    import json
    import socket
    import subprocess
    env = dict(os.environ, VBQ_BENCH_ONE_DEVICE="1", VBQ_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    full = str(tmp_path / "bench_full.json")
    tail = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-other-workloads", "--full-record", full] + extra
    if launcher == "torchrun":                    # as the driver launches N > 1
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + tail
    else:                                         # plain `python bench.py --gpus 2`: bench.py starts its own ranks
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len([l for l in lines if l.startswith("{")]) == 1 and lines[-1].startswith("{")
    assert len(lines[-1]) < 4096                  # what the driver's parser takes
    return json.loads(lines[-1]), json.load(open(full))


def _check_compact_line(line, d):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity_vs_oracle_on_sample"):
        assert k in line, k
    assert line["value"] == d["value"] and line["ms_per_step"] == d["ms_per_step"] and line["n_gpus"] == d["n_gpus"]
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
    assert line["parity_vs_oracle_on_sample"] is True and len(line["per_gpu"]) == 2
    # the N > 1 line is self-sufficient: the CPU path timed in the same run (rank 0's bounded sample), all three stage fractions
    cb = line["cpu_baseline"]
    assert cb is not None and cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and "rank 0 of 2" in d["cpu_baseline"]["sample"]
    assert set(line["stages_frac"]) == {"k1t", "k1", "k2"} and all(0 < v < 1 for v in line["stages_frac"].values())


@pytest.mark.timeout(900)
def test_bench_two_ranks_code_path(tmp_path):
    """bench.py's N > 1 path end to end as the driver launches it (torch.distributed.run, 2 ranks), on one GPU with gloo
    as the transport (VBQ_BENCH_ONE_DEVICE / VBQ_BENCH_BACKEND: testing switches): one JSON line from rank 0 with the
    all-reduce report, the packed-counter guard passing, twice the single-rank work accounted for -- and CORRECTNESS
    evidence: one code book on both ranks, each rank's indices against the oracle, the all-reduced histograms against the
    sum of the ranks' single-launch histograms, the R-D curve against the oracle's, every rank's own kernel report."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    line, d = _run_bench_two_ranks(["--workload", "kodak24_c32"], tmp_path)
    _check_compact_line(line, d)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    ar = d["allreduce"]
    assert ar["packed_3x21"] is True and ar["rank_histogram_payload_bytes"] == ((32 * 32 * 2047 + 2) // 3 + 1) * 8
    assert ar["rank_histogram_allreduce_ms_isolated"] > 0 and ar["ms_per_step_without_collectives"] > 0
    assert ar["reserved_workgroups"] == 64            # 5.6 MB beside the next step's kernels: slots are left for the collective
    assert d["config"]["elements_per_gpu"] == 36864 * 32
    assert d["pairs_per_step"] == 2 * 36864 * 32 * 32                      # both ranks' pairs, counted once
    assert d["value"] == pytest.approx(d["pairs_per_step"] / (d["ms_per_step"] * 1e-3), rel=1e-6)
    par = d["parity"]
    assert d["parity_vs_oracle_on_sample"] is True and [f["ok"] for f in par["ranks"]] == [True, True]
    for key in ("pass2_indices_equal_oracle_on_windows", "pass1_level_counts_equal_k1_k2_route_summed_over_ranks",
                "pass2_rank_counts_equal_single_launch_summed_over_ranks", "counts_total", "length_table_equals_numpy",
                "models_equal_numpy", "rd_curve_within_1e-5_of_oracle"):
        assert par[key] is True, key
    pg = d["per_gpu"]
    assert [g["rank"] for g in pg] == [0, 1] and all(g["pass2_k1_ms"] > 0 and 0 < g["k1_hbm_frac"] < 1 for g in pg)
    assert all(g["pass2_k1_ms_without_collectives"] > 0 and g["pass1_k1t_ms_without_collectives"] > 0 for g in pg)
    assert len(d["rd_curve"]["lagrangian_per_latent"]) == 32 and d["rd_curve"]["vs_oracle_on_sample"]["max_rel_diff"] <= 1e-5


@pytest.mark.timeout(900)
def test_bench_two_ranks_self_launched(tmp_path):
    """Plain `python bench.py --gpus 2` (no torch.distributed.run around it): the parent, which makes no GPU call, starts
    the two ranks itself and relays rank 0's compact line as its last line."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    line, d = _run_bench_two_ranks(["--workload", "kodak24_c32"], tmp_path, launcher="self")
    _check_compact_line(line, d)
    assert d["n_gpus"] == 2 and d["pairs_per_step"] == 2 * 36864 * 32 * 32
    assert [f["ok"] for f in d["parity"]["ranks"]] == [True, True]


@pytest.mark.timeout(900)
def test_bench_two_ranks_strong_scaling_splits_one_tensor(tmp_path):
    """--scaling strong: the rows of ONE tensor are split over the ranks (BASELINE configs[3]'s curve), the pairs of the
    whole tensor are counted once, and the global histograms hold exactly that tensor."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    _, d = _run_bench_two_ranks(["--workload", "kodak24_c32", "--scaling", "strong"], tmp_path)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["elements_per_gpu"] == 18432 * 32 and d["pairs_per_step"] == 36864 * 32 * 32
    assert d["parity_vs_oracle_on_sample"] is True and d["parity"]["counts_total"] is True


# ---------------------------------------------------------------------------------------------------------------
# The stream-ordered HOST STAGE with world > 1: every large C = 1 configuration of BASELINE.json (1e8, 1.2e8, the 1.25e8
# shards of the 1e9 run) has 2^24 rows per histogram row or more, so its -log2 steps cannot be tabulated and run as NumPy
# on the HIP runtime's callback thread (pipeline.HostStage) -- beside the asynchronous, double-buffered all-reduces.
HS_ROWS = (1 << 24) + 4096 + 7                 # global rows: just above the tabulation limit, ragged
HS_LAMBS = [0.05, 4.0]


def _host_stage_data():
    from scipy.stats import norm
    rng = np.random.default_rng(2024)
    mu = rng.standard_normal(HS_ROWS, dtype=np.float32) * np.float32(1.23) - np.float32(0.08)
    sg = np.exp(rng.standard_normal(HS_ROWS, dtype=np.float32) * np.float32(0.7) - np.float32(2.0))
    xi = np.concatenate([(np.arange(2 ** n) + 0.5) / 2 ** n for n in range(N + 1)])
    tab = norm.ppf(xi[None, :], scale=1.2329).astype(np.float32)
    return mu, sg, tab


def _host_stage_build(mu, sg, tab, steps, **kw):
    from vbq_amd.pipeline import EntropyModelBuild
    dev = torch.device("cuda", torch.cuda.current_device())
    mu_d, sg_d = torch.from_numpy(mu).to(dev).reshape(1, -1), torch.from_numpy(sg).to(dev).reshape(1, -1)
    b = EntropyModelBuild(mu.size, 1, HS_LAMBS, torch.from_numpy(tab).to(dev), N=N, add_n_smoothing=1, **kw)
    assert b.has_host_stages and b.lut1 is None and b.lut2 is None, "the test must take the host-stage route"
    assert b.host_stage_kind == ("python" if os.environ.get("VBQ_PYTHON_HOST_STAGE") == "1" else "native")
    for _ in range(steps):                        # several steps: both histogram buffers, the model table one step late
        b.run(mu_d, sg_d)
    models = b.finish_models()
    b.check()
    torch.cuda.synchronize()
    return (b.level_counts.cpu().numpy(), b.level_len.cpu().numpy(), b.raw_models.cpu().numpy(),
            b.counts.cpu().numpy().astype(np.int64), models.cpu().numpy())


def _host_stage_worker(rank, world, port, out, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from vbq_amd import dist as vd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if backend == "nccl":                                      # real GPUs, RCCL (tests/test_zz_multi_gpu.py)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
    mu, sg, tab = _host_stage_data()
    a, b = vd.shard_rows(HS_ROWS, rank, world)
    res = _host_stage_build(mu[a:b], sg[a:b], tab, 3, global_rows=HS_ROWS, distributed=True, group=dist.group.WORLD,
                            level_group=dist.new_group())
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_build_through_the_host_stage_equals_single_process():
    """Two ranks (one device, gloo) x three steps of a C = 1 build with global_rows >= 2^24: both -log2 steps run as NumPy on the
    HIP runtime's callback thread (as plain C around NumPy's own log2 loop; the Python form gives the same tables), stream-ordered between the kernels and beside the asynchronous all-reduces (both histogram
    buffers in use, the model table of a step looked up one step late) -- and every rank ends with exactly the histograms,
    length table and models one process computes from all rows, which in turn are the reference's NumPy arithmetic on the
    oracle's counts."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a ROCm device")
    from oracle import c_oracle as CO, vbq_oracle as O
    from vbq_amd import entropy
    mu, sg, tab = _host_stage_data()
    ref = _host_stage_build(mu, sg, tab, 2)
    # plain C with NumPy's log2 loop (no interpreter lock on the callback thread) == NumPy in Python on that thread
    os.environ["VBQ_PYTHON_HOST_STAGE"] = "1"
    try:
        ref_py = _host_stage_build(mu, sg, tab, 2)
    finally:
        del os.environ["VBQ_PYTHON_HOST_STAGE"]
    assert all(np.array_equal(a, b) for a, b in zip(ref, ref_py))
    # the single-process build itself against the oracle: level histogram of pass 1, models = NumPy float32 ops on the counts
    lev = O.levels_of_sorted_ranks(N)
    n = 300_000
    w1 = CO.quantize(mu[:n, None], sg[:n, None], tab, HS_LAMBS, N=N)
    from vbq_amd import ops
    lc_n = ops.level_counts(torch.from_numpy(mu[:n]).cuda(), torch.from_numpy(sg[:n]).cuda(), torch.from_numpy(tab).cuda(), HS_LAMBS, N=N)
    assert np.array_equal(lc_n.cpu().numpy()[:, 0], np.stack([np.bincount(lev[w1[l, :, 0]], minlength=N + 1) for l in range(2)]))
    assert int(ref[0].sum()) == 2 * HS_ROWS and int(ref[3].sum()) == 2 * HS_ROWS
    lv = np.arange(N + 1, dtype=np.float32)
    assert np.array_equal(ref[2], entropy.neg_log2_freq(ref[0], 1)) and np.array_equal(ref[1], (lv + ref[2]).astype(np.float32))
    assert np.array_equal(ref[4], entropy.neg_log2_freq(ref[3], 1))
    w2 = CO.quantize(mu[:n, None], sg[:n, None], tab, HS_LAMBS, N=N, level_len=ref[1])
    idx_n = ops.quantize(torch.from_numpy(mu[:n]).cuda(), torch.from_numpy(sg[:n]).cuda(), torch.from_numpy(tab).cuda(), HS_LAMBS, N=N,
                         level_len=torch.from_numpy(ref[1]).cuda())
    assert np.array_equal(idx_n.cpu().numpy(), w2[:, :, 0])
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29100 + os.getpid() % 150
    procs = [ctx.Process(target=_host_stage_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=500) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(2):
        for got, want, what in zip(res[r], ref, ("level_counts", "level_len", "raw_models", "counts", "models")):
            assert np.array_equal(got, want), (r, what)

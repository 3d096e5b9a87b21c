"""The N > 1 path on CPU: two gloo ranks shard the rows, all-reduce their histograms and
moments, and must arrive at the entropy models a single process computes."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 6
T = 2 ** (N + 1) - 1


def _make(seed=0, rows=501, C=3, L=4):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, T, (L, rows, C)).astype(np.int64)
    x = rng.normal(0, 1.3, (rows, C))
    return idx, x


def _counts(idx, C):
    L = idx.shape[0]
    out = np.zeros((L, C, T), np.int64)
    for l in range(L):
        for c in range(C):
            out[l, c] = np.bincount(idx[l, :, c], minlength=T)
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from vbq_amd import dist as vd
    r, w, dev = vd.init_from_env("gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    idx, x = _make()
    a, b = vd.shard_rows(idx.shape[1], rank, world)
    local = torch.from_numpy(_counts(idx[:, a:b], idx.shape[2]))
    raw, full = vd.entropy_models_from_local_counts(local, N, 1)
    xs = x[a:b]
    mom = torch.from_numpy(np.stack([xs.sum(0), (xs ** 2).sum(0)], axis=1))
    std = vd.global_empirical_std(mom, xs.shape[0])
    # the asynchronous int32 reducer bench.py uses (on CPU it reduces the int32 tensor as it is; the packed
    # 3 x 21-bit form needs the device kernels and is covered by tests/test_gpu_dist.py)
    c32 = local.to(torch.int32).contiguous()
    red = vd.CountsAllReduce(c32.numel(), dev, max_global_count=idx.shape[1])
    assert red.packed is False
    red.start(c32).wait()
    assert torch.equal(c32.to(torch.int64), torch.from_numpy(_counts(idx, idx.shape[2])))
    q.put((rank, raw, full, std, (a, b)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_histogram_allreduce_matches_single_process(world):
    """world = 2 and an odd world size (three ranks, shards of 167 rows): integer sums make the models independent of the
    number of ranks."""
    from vbq_amd import entropy
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    idx, x = _make()
    g = torch.from_numpy(_counts(idx, idx.shape[2]))
    raw_ref = entropy.neg_log2_freq(entropy.level_counts_from_counts(g, N), 1)
    full_ref = entropy.neg_log2_freq(g, 1)
    spans = [r[4] for r in res]
    assert spans[0][0] == 0 and spans[-1][1] == idx.shape[1] and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    for _, raw, full, std, _ in res:
        assert np.array_equal(raw, raw_ref) and np.array_equal(full, full_ref)       # bit-identical on every rank
        assert np.allclose(std, np.sqrt((x ** 2).mean(0)), rtol=1e-12)


def test_shard_rows_covers_everything():
    from vbq_amd.dist import shard_rows
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            spans = [shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1

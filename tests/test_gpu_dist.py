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

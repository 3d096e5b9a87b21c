#!/usr/bin/env python3
"""Developer tool: ways to land the three [16, 1, 32, 48, 256] float32 results of one image (3 x 25 MB) in fresh NumPy arrays."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda")
shape = (16, 1, 32, 48, 256)
src = [torch.randn(shape, device=dev) for _ in range(3)]
pin = [torch.empty(shape, dtype=torch.float32, pin_memory=True) for _ in range(3)]
pool3, pool6, pool9 = ThreadPoolExecutor(3), ThreadPoolExecutor(6), ThreadPoolExecutor(9)
cuts = [0, 5, 10, 16]


def issue(q, pieces):
    evs = []
    for a, b in (zip(cuts[:-1], cuts[1:]) if pieces else [(0, 16)]):
        pin[q][a:b].copy_(src[q][a:b], non_blocking=True)
        e = torch.cuda.Event()
        e.record()
        evs.append((a, b, e))
    return evs


def copy_piece(out, q, a, b, e):
    e.synchronize()
    np.copyto(out[a:b], pin[q].numpy()[a:b])


def v_seq_pieces():            # per quantity: 3 piece DMAs, 3 threads copy out; next quantity afterwards
    res = []
    for q in range(3):
        evs = issue(q, True)
        out = np.empty(shape, np.float32)
        list(pool3.map(lambda t: copy_piece(out, q, *t), evs))
        res.append(out)
    return res


def v_all_then_pieces(pool=pool3):     # all DMAs up front; per quantity 3 threads
    evs = [issue(q, True) for q in range(3)]
    res = []
    for q in range(3):
        out = np.empty(shape, np.float32)
        list(pool.map(lambda t: copy_piece(out, q, *t), evs[q]))
        res.append(out)
    return res


def v_all_nine(pool):          # all DMAs up front; nine (quantity, piece) tasks
    evs = [issue(q, True) for q in range(3)]
    outs = [np.empty(shape, np.float32) for _ in range(3)]
    list(pool.map(lambda qt: copy_piece(outs[qt[0]], qt[0], *qt[1]), [(q, t) for q in range(3) for t in evs[q]]))
    return outs


def v_eager_r4():              # all DMAs up front (whole quantities); one thread per quantity, np.array of the staging view
    evs = [issue(q, False)[0] for q in range(3)]

    def one(q):
        evs[q][2].synchronize()
        return np.array(pin[q].numpy())
    return list(pool3.map(one, range(3)))


def v_torch_cpu():
    return [s.cpu().numpy() for s in src]


keep = [np.empty(shape, np.float32) for _ in range(3)]


def v_reuse_outputs():         # no fresh allocation: what page faults cost
    evs = [issue(q, True) for q in range(3)]
    list(pool9.map(lambda qt: copy_piece(keep[qt[0]], qt[0], *qt[1]), [(q, t) for q in range(3) for t in evs[q]]))
    return keep


def v_dma_only():
    for q in range(3):
        issue(q, False)
    torch.cuda.synchronize()


def timeit(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ref = [s.cpu().numpy() for s in src]
for name, f in (("DMA only (75 MB into pinned)", v_dma_only), ("per quantity: pieces + 3 threads", v_seq_pieces),
                ("all DMAs first, per quantity 3 threads", v_all_then_pieces), ("all DMAs first, nine tasks / 3 threads", lambda: v_all_nine(pool3)),
                ("all DMAs first, nine tasks / 6 threads", lambda: v_all_nine(pool6)), ("all DMAs first, nine tasks / 9 threads", lambda: v_all_nine(pool9)),
                ("eager r4: one thread per quantity", v_eager_r4), ("torch .cpu() x 3", v_torch_cpu), ("nine tasks into kept outputs", v_reuse_outputs)):
    ms = timeit(f)
    r = f()
    ok = r is None or all(np.array_equal(a, b) for a, b in zip(r, ref))
    print(f"{name:45s} {ms:7.3f} ms  ok={ok}")

# ---- the same through the quantizer: compress_latents with NumPy in / out (lazy views), everything read / num_bits alone
import vbq_amd
from bench import LAMBDAS_16, N_BITS, make_inputs_with_table

rows, C = 36864, 256
mu_h, sg_h, tab_h = make_inputs_with_table(rows, C, 1000)
mu_bc, sg_bc = torch.from_numpy(mu_h).to(dev), torch.from_numpy(sg_h).to(dev)


class _Table:
    def inverse_cdf(self, xi):
        return np.ascontiguousarray(tab_h.T)


q = vbq_amd.ChannelwisePriorCDFQuantizer(C, N_BITS)
q.build_code_points(_Table())
q.build_entropy_models_from_latents(mu_bc, sg_bc, LAMBDAS_16, 1)
m_np = mu_h[:1536].reshape(1, 32, 48, C).copy()
lv_np = (2 * np.log(sg_h[:1536])).astype(np.float32).reshape(1, 32, 48, C)
m_d, lv_d = torch.from_numpy(m_np).to(dev), torch.from_numpy(lv_np).to(dev)


def call(read, device_in=False):
    o = q.compress_latents(m_d if device_in else m_np, lv_d if device_in else lv_np, LAMBDAS_16)
    for k in read:
        np.asarray(o[k][LAMBDAS_16[0]])
    return None


for name, f in (("compress_latents NumPy in, nothing read", lambda: call(())),
                ("compress_latents device in, nothing read", lambda: call((), True)),
                ("compress_latents NumPy in, all three read", lambda: call(("Z_hat", "raw_num_bits", "num_bits"))),
                ("compress_latents device in, all three read", lambda: call(("Z_hat", "raw_num_bits", "num_bits"), True)),
                ("compress_latents NumPy in, num_bits read", lambda: call(("num_bits",))),
                ("compress_latents device in, num_bits read", lambda: call(("num_bits",), True)),
                ("eager r4 landing alone (again)", v_eager_r4)):
    print(f"{name:45s} {timeit(f):7.3f} ms")

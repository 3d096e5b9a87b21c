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

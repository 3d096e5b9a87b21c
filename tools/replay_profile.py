#!/usr/bin/env python3
"""Developer tool: where one replayed image of the evaluation loop spends its time (device: the graph alone between events;
host: compress_replay's Python around it).   python tools/replay_profile.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vbq_amd
from bench import LAMBDAS_16, N_BITS, make_inputs, empirical_tables

dev = torch.device("cuda")
rows, C = 36864, 256
mu_h, sg_h = make_inputs(rows, C, seed=1000)
mu_bc, sg_bc = torch.from_numpy(mu_h).to(dev), torch.from_numpy(sg_h).to(dev)
tab_h = empirical_tables(mu_bc, rows, C, "bc")


class _Table:
    def inverse_cdf(self, xi):
        return np.ascontiguousarray(tab_h.T)


q = vbq_amd.ChannelwisePriorCDFQuantizer(C, N_BITS)
q.build_code_points(_Table())
lams = LAMBDAS_16
q.build_entropy_models_from_latents(mu_bc, sg_bc, lams, 1)
H, W = 32, 48
B = H * W
m_img = mu_bc[:B].reshape(1, H, W, C).contiguous()
lv_img = (2.0 * torch.log(sg_bc[:B])).reshape(1, H, W, C).contiguous()


class VAE:
    def encode(self, X):
        return m_img, lv_img

    def decode(self, Z):
        return 0.5 + 0.1 * Z[..., :3]


vae = VAE()
X = np.zeros((1, H, W, 3), np.float32)
for _ in range(20):
    q.compress_replay(X, vae, lams)
rp = next(iter(q._dev_cache["_replays"].values()))
print("mode", rp.mode, rp.errors)
n = 300
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    q.compress_replay(X, vae, lams)
t_all = (time.perf_counter() - t0) / n * 1e3
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for a, b in ev:
    a.record(); rp.graph.replay(); b.record()
torch.cuda.synchronize()
t_graph = float(np.median([a.elapsed_time(b) for a, b in ev]))
t0 = time.perf_counter()
for _ in range(n):
    q._replay_fingerprint(lams)
t_fp = (time.perf_counter() - t0) / n * 1e3
t0 = time.perf_counter()
for _ in range(n):
    rp._prepare(rp.mode, X)
torch.cuda.synchronize()
t_prep = (time.perf_counter() - t0) / n * 1e3
t0 = time.perf_counter()
for _ in range(n):
    rp.graph.replay()
    torch.cuda.current_stream().synchronize()
t_rs = (time.perf_counter() - t0) / n * 1e3
t0 = time.perf_counter()
for _ in range(n):
    tuple(np.array(h.numpy()) for h in rp._host_reads)
t_np = (time.perf_counter() - t0) / n * 1e3
print(f"compress_replay per image {t_all:.4f} ms; graph alone (events) {t_graph:.4f}; replay + sync (host clock) {t_rs:.4f}; "
      f"fingerprint {t_fp:.4f}; prepare (copy X in) {t_prep:.4f}; host copies of the reads {t_np:.4f}")

import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import LAMBDAS, N_BITS, make_inputs_with_table
from vbq_amd import ops, _lib
dev = torch.device("cuda")
rows, C = 36864, 256
mu_h, sg_h, tab_h = make_inputs_with_table(rows, C, 1000)
mu = torch.from_numpy(np.ascontiguousarray(mu_h.T)).to(dev); sg = torch.from_numpy(np.ascontiguousarray(sg_h.T)).to(dev)
tab = torch.from_numpy(tab_h).to(dev)
rng = np.random.default_rng(5)
ll = torch.from_numpy((np.arange(N_BITS + 1, dtype=np.float32)[None, None, :] + np.abs(rng.normal(0, 1.0, (32, C, N_BITS + 1)))).astype(np.float32)).to(dev)
idx = torch.empty((32, C, rows), dtype=torch.uint16, device=dev)
f = lambda: ops.quantize(mu, sg, tab, LAMBDAS, N=N_BITS, layout="cb", out_idx=idx, level_len=ll)
for _ in range(300): f()
torch.cuda.synchronize()
f(); torch.cuda.synchronize()
h = _lib.lib()
n = 2 * 1024
buf = (ctypes.c_ulonglong * n)()
h.vbq_debug_k1_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", h.vbq_debug_k1_stamps(buf, n))
a = np.array(buf, dtype=np.uint64).reshape(-1, 2).astype(np.int64)
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0     # s_memrealtime ticks at 100 MHz -> us
print("workgroups", len(a), "start us: min %.1f p50 %.1f max %.1f" % (st.min(), np.median(st), st.max()))
print("end us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
dur = en - st
print("duration us: min %.1f p50 %.1f max %.1f" % (dur.min(), np.median(dur), dur.max()))
hist, edges = np.histogram(en, bins=12)
print("end-time histogram:", list(zip(np.round(edges[:-1]).astype(int), hist)))

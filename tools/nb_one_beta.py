import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from bench import make_inputs, N_BITS, BETAS_50
from vbq_amd import ops, embeddings as Emb
from oracle import c_oracle as CO, vbq_oracle as O
from tools.abtime import timeit
dev = torch.device("cuda")
n = 10_000_000
m_h, s_h = make_inputs(n, 1, 1000)
m, s = torch.from_numpy(m_h.reshape(n)).to(dev), torch.from_numpy(s_h.reshape(n)).to(dev)
pts_h, lens_h = Emb.make_code_book(Emb.empirical_std(m), N_BITS)
cb = torch.from_numpy(pts_h).to(dev)
r2s = O.level_major_to_rank(N_BITS)
nn = 200000
for b in (BETAS_50[0], BETAS_50[10], BETAS_50[25], BETAS_50[40], BETAS_50[49]):
    ix = torch.empty((1, n), dtype=torch.uint16, device=dev)
    med, best = timeit(lambda: ops.quantize_notebook(m, s, cb, [b], N=N_BITS, want_values=False, out_idx=ix))
    want = r2s[CO.compress_coordinates(m_h[:nn, 0], s_h[:nn, 0], b, pts_h, lens_h, threads=CO.max_threads())[1]]
    ok = np.array_equal(ix[0, :nn].cpu().numpy().astype(np.int64), want)
    i2, v2 = ops.quantize_notebook(m[:nn+1], s[:nn+1], cb, [b, BETAS_50[30]], N=N_BITS, want_values=True)
    want2 = CO.compress_coordinates(m_h[:nn+1, 0], s_h[:nn+1, 0], b, pts_h, lens_h, threads=CO.max_threads())
    ok2 = np.array_equal(i2[0].cpu().numpy().astype(np.int64), r2s[want2[1]]) and np.array_equal(v2[0].cpu().numpy(), want2[0])
    print(f"{os.environ.get('VBQ_NO_PRUNED','0')} beta {b:10.4g}: {med*1e3:7.1f} us  {n/med/1e6:.1f} G/s  idx ok {ok}  pair+values ok {ok2}", flush=True)

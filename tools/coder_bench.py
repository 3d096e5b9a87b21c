"""Throughput of the rANS coder on the Kodak-24 sweep (3.0e8 symbols in 8192 streams)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import LAMBDAS, N_BITS, make_inputs_with_table as make_inputs
from vbq_amd import ops
from vbq_amd.coder import RansCodec, quantize_frequencies, ideal_bits
from tools.kbench import timeit
dev = torch.device("cuda")
rows, C = 36864, 256
mu, sg, tab = make_inputs(rows, C, 0)
mu_t, sg_t = (torch.from_numpy(np.ascontiguousarray(a.T)).to(dev) for a in (mu, sg))
idx = ops.quantize(mu_t, sg_t, torch.from_numpy(tab).to(dev), LAMBDAS, N=N_BITS, layout="cb")
counts = ops.histogram(idx, C, N=N_BITS, layout="cb")
freq = quantize_frequencies(counts)
for seg in (256, 512, 576, 1024, 4096):
    codec = RansCodec(freq.reshape(-1, 2047), N=N_BITS, segment=seg)
    words, sizes = codec.encode(idx)
    e_med, _ = timeit(lambda: codec.encode(idx), 5)
    d_med, _ = timeit(lambda: codec.decode(words, sizes, rows), 5)
    bits = codec.compressed_bits(sizes)
    est = ideal_bits(counts, freq)
    n = idx.numel()
    print(f"segment {seg}: encode {e_med:.2f} ms ({n/e_med/1e6:.1f} G sym/s), decode {d_med:.2f} ms ({n/d_med/1e6:.1f} G sym/s), "
          f"{bits/n:.4f} bits/sym vs cross-entropy {est/n:.4f} (+{100*(bits/est-1):.2f} %)")

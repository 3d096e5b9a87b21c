"""Developer probe: is HIP's expf the same function as torch.exp on this device, for EVERY float32?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbq_amd import _lib, ops
h = _lib.lib()
dev = torch.device("cuda")
n = 1 << 26
bad_total = 0
first = []
for chunk in range(64):
    bits = torch.arange(chunk * n, (chunk + 1) * n, dtype=torch.int64, device=dev).to(torch.int32) if chunk < 32 else \
        (torch.arange(chunk * n, (chunk + 1) * n, dtype=torch.int64, device=dev) - (1 << 32)).to(torch.int32)
    x = bits.view(torch.float32).reshape(n // 256, 256)
    want = torch.exp(x) ** 0.5
    mu = torch.empty((256, n // 256), dtype=torch.float32, device=dev)
    sg = torch.empty_like(mu)
    _lib.check(h.vbq_prep_planes_f32(ops._ptr(x), ops._ptr(x), 2, n // 256, 256, ops._ptr(mu), ops._ptr(sg), ops._stream(x)), "prep")
    got = sg.t()
    ne = (got.view(torch.int32) != want.view(torch.int32)) & ~(torch.isnan(got) & torch.isnan(want))
    k = int(ne.sum())
    bad_total += k
    if k and len(first) < 8:
        i = ne.nonzero()[:3]
        for r, c in i.tolist():
            first.append((float(x[r, c]), float(got[r, c]), float(want[r, c])))
print("values differing from torch.exp(x) ** 0.5 over all 2^32 inputs:", bad_total)
for f in first:
    print("   x, mine, torch:", f)

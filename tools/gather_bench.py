#!/usr/bin/env python3
"""Developer tool: vbq_gather_latents_u16 on one Kodak image (16 lambdas) and on the whole Kodak-24 tensor (32 lambdas)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.abtime import timeit
from vbq_amd import ops

dev = torch.device("cuda")
T = 2047
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="image,kodak24")
args = ap.parse_args()
SIZES = {"image": (16, 256, 1536), "kodak24": (32, 256, 36864), "2img": (16, 256, 3072), "4img": (16, 256, 6144), "8img": (16, 256, 12288),
         "16img": (16, 256, 24576), "kodak24_L16": (16, 256, 36864), "4img_L32": (32, 256, 6144), "img_L32": (32, 256, 1536)}
for name, (L, C, B) in ((k, SIZES[k]) for k in args.sizes.split(",")):
    g = torch.Generator(device=dev).manual_seed(1)
    idx = torch.randint(0, T, (L, C, B), device=dev, generator=g, dtype=torch.int32).to(torch.uint16)
    srt = torch.sort(torch.randn((C, T), device=dev, generator=g), dim=1).values.contiguous()
    ll = torch.rand((L, C, 11), device=dev, generator=g) * 20
    md = torch.rand((L, C, T), device=dev, generator=g) * 15
    for what, kw in (("z", dict(want_raw_bits=False)), ("z+raw+nb", dict(level_len=ll, models=md, want_num_bits=True)),
                     ("nb", dict(want_zhat=False, want_raw_bits=False, models=md, want_num_bits=True)),
                     ("z+idx", dict(want_raw_bits=False, want_idx=True)), ("idx", dict(want_zhat=False, want_raw_bits=False, want_idx=True))):
        med, mn = timeit(lambda: ops.gather_latents(idx, N=10, table_sorted=srt, **kw))
        n_out = {"z": 4, "z+raw+nb": 12, "nb": 4, "z+idx": 6, "idx": 2}[what]
        byt = L * C * B * (2 + n_out)
        print(f"{name:8s} {what:9s} {med * 1e3:8.1f} us (min {mn * 1e3:.1f})  {byt / med / 1e9:6.2f} TB/s on {byt / 1e6:.0f} MB")
    med, mn = timeit(lambda: ops.transpose_planes(idx))
    print(f"{name:8s} transpose_planes u16 {med * 1e3:8.1f} us  {L * C * B * 4 / med / 1e9:6.2f} TB/s")

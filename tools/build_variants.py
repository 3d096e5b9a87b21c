#!/usr/bin/env python3
"""Developer tool: A/B variants of libvbq_hip.so for ONE gpurun call.

    python tools/build_variants.py tag1:-DFOO=1,-DBAR tag2:-DFOO=2 ...

Recompiles the sources that read experiment macros (vbq_quantize_fast.hip, vbq_notebook.hip, vbq_hist.hip) with the extra
flags of each tag, links them with the cached objects of the regular build and writes tools/bin/libvbq_<tag>.so
(git-ignored; *.so files travel to the GPU box).  Select one with VBQ_HIP_LIBRARY=tools/bin/libvbq_<tag>.so.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vbq_amd import build as B  # noqa: E402

VARIANT_SOURCES = ["vbq_quantize_fast.hip", "vbq_notebook.hip", "vbq_hist.hip", "vbq_latents.hip"]


def one(tag, flags):
    objdir = os.path.join(B.LIBDIR, "obj")
    vdir = os.path.join(ROOT, "tools", "bin", "obj_" + tag)
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for s in B.HIP_SOURCES:
        base = s[:-4] + ".o"
        if s in VARIANT_SOURCES:
            o = os.path.join(vdir, base)
            B._compile_one("hipcc", os.path.join(B.CSRC, s), o, flags + B.extra_flags())
            objs.append(o)
        else:
            objs.append(os.path.join(objdir, base))
    out = os.path.join(ROOT, "tools", "bin", f"libvbq_{tag}.so")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + B.LINK_LIBS + ["-o", out], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr)
    return out


def main():
    specs = []
    for a in sys.argv[1:]:
        tag, _, fl = a.partition(":")
        specs.append((tag, [f for f in fl.split(",") if f]))
    B.build_hip()
    with ThreadPoolExecutor(max_workers=4) as ex:
        for out in ex.map(lambda s: one(*s), specs):
            print(out)


if __name__ == "__main__":
    main()

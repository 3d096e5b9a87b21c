"""Build the HIP extension (and, for tests only, the C oracle) in-tree.

    python -m vbq_amd.build            # libvbq_hip.so for gfx950
    python -m vbq_amd.build --oracle   # also oracle/_build/libvbq_oracle.so

hipcc cross-compiles gfx950 without a GPU; the resulting .so is git-ignored but travels
with the working tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libvbq_hip.so")
INCLUDE = os.path.join(ROOT, "include")
HIP_SOURCES = ["vbq_api.hip", "vbq_quantize.hip", "vbq_quantize_fast.hip", "vbq_hist.hip", "vbq_notebook.hip", "vbq_bmshj.hip",
               "vbq_candidates.hip", "vbq_latents.hip", "vbq_host.hip", "vbq_rans.hip", "vbq_comm.hip", "vbq_xi.hip", "vbq_baselines.hip", "vbq_ranks.hip", "vbq_metrics.hip"]
LINK_LIBS = ["-ldl"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
               # every f32/f64 op of the reference is a separately rounded op: never contract a*b+c
               "-ffp-contract=off", "-fno-fast-math",
               # no SLP vectorisation: the v_pk_add/mul_f32 pairs it forms issue at the rate of two scalar ops but need
               # their operands in aligned register pairs -- the moves cost K1t 3 %, K1e 5 %, the one-lambda kernel 7 %
               "-fno-slp-vectorize"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def hip_sources():
    return [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _tmp(path):
    # several ranks (or threads) may find a stale library at the same time: every builder writes its own temporary
    # file and installs it with an atomic rename, so a reader never sees a half-written object or library
    import threading
    return f"{path}.{os.getpid()}.{threading.get_ident()}.tmp"


def _compile_one(hipcc, src, obj, extra):
    tmp = _tmp(obj)
    cmd = [hipcc] + HIPCC_FLAGS + extra + ["-I", INCLUDE, "-I", CSRC, "-c", src, "-o", tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on " + os.path.basename(src) + ":\n" + r.stdout + r.stderr)
    os.replace(tmp, obj)
    return " ".join(cmd)


def extra_flags():
    extra = os.environ.get("VBQ_EXTRA_HIPCC_FLAGS", "").split()       # developer experiments only
    if os.environ.get("VBQ_ONLY_N10") == "1":   # the reference's bit depth alone (post_process.py:117): a third of the build time
        extra.append("-DVBQ_ONLY_N10")
    return extra


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """One object per .hip source (rebuilt only when it or a header changed, in parallel), one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = hip_sources()
    headers = [os.path.join(CSRC, "vbq_common.h"), os.path.join(INCLUDE, "vbq.h")]
    extra = extra_flags()
    objdir = os.path.join(LIBDIR, "obj")
    flags_tag = os.path.join(objdir, "flags.txt")
    lib_tag = os.path.join(LIBDIR, "flags.txt")                 # travels with the library (the object cache does not)
    tag = " ".join(HIPCC_FLAGS + extra)
    # A tree that arrived with its library but without the object cache (the GPU box: vbq_amd/lib/obj is not shipped) is up
    # to date when the library is newer than every source AND was built with these flags: nothing to do, and nothing
    # that needs hipcc.
    if not force and not os.path.isdir(objdir) and not _newer(LIB, srcs + headers) and \
            os.path.exists(lib_tag) and open(lib_tag).read() == tag:
        return LIB
    os.makedirs(objdir, exist_ok=True)
    if not os.path.exists(flags_tag) or open(flags_tag).read() != tag:
        force = True
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    todo = [(s, o) for s, o in zip(srcs, objs) if force or _newer(o, [s] + headers)]
    if not todo and not _newer(LIB, objs) and os.path.exists(lib_tag) and open(lib_tag).read() == tag:
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libvbq_hip.so (ROCm toolchain required)")
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for line in ex.map(lambda so: _compile_one(hipcc, so[0], so[1], extra), todo):
            if verbose:
                print(line)
    open(flags_tag, "w").write(tag)
    tmp = _tmp(LIB)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + LINK_LIBS + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    os.replace(tmp, LIB)
    tmp = _tmp(lib_tag)
    open(tmp, "w").write(tag)
    os.replace(tmp, lib_tag)
    return LIB


def build_oracle(force: bool = False) -> str:
    """gcc build of oracle/vbq_oracle.c (test infrastructure, never loaded by vbq_amd)."""
    odir = os.path.join(ROOT, "oracle")
    r = subprocess.run(["make", "-C", odir] + (["-B"] if force else []), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return os.path.join(odir, "_build", "libvbq_oracle.so")


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
    if "--oracle" in sys.argv:
        print(build_oracle(force="--force" in sys.argv))

"""ctypes wrapper of oracle/_build/libvbq_oracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libvbq_oracle.so")
_H = None


def lib(build=True):
    global _H
    if _H is None:
        if build and (not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "vbq_oracle.c"))):
            subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
        _H = C.CDLL(SO)
    return _H


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def max_threads():
    return lib().vbq_oracle_max_threads()


def quantize(mu, sigma, table_lm, lambdas, N=10, level_len=None, layout=0, mode=0, want_zhat=False,
             want_bits=False, threads=1):
    """mu, sigma: [rows, C] (layout 0) or [C, rows] (layout 1) f32.  Returns idx u16 [L, *mu.shape] (+ zhat, bits)."""
    mu = np.ascontiguousarray(mu, np.float32)
    sigma = np.ascontiguousarray(sigma, np.float32)
    if mu.ndim == 1:
        mu, sigma = mu[:, None], sigma[:, None]
    rows, Cc = (mu.shape if layout == 0 else mu.shape[::-1])
    table_lm = np.ascontiguousarray(table_lm, np.float32).reshape(Cc, -1)
    lam = np.ascontiguousarray(lambdas, np.float64)
    L = lam.shape[0]
    ll = None if level_len is None else np.ascontiguousarray(level_len, np.float32)
    idx = np.empty((L,) + mu.shape, np.uint16)
    zh = np.empty((L,) + mu.shape, np.float32) if want_zhat else None
    bt = np.empty((L,) + mu.shape, np.float32) if want_bits else None
    r = lib().vbq_oracle_quantize_f32(_p(mu), _p(sigma), C.c_int64(rows), C.c_int32(Cc), C.c_int32(layout),
                                      _p(table_lm), _p(ll), _p(lam), C.c_int32(L), C.c_int32(N), C.c_int32(mode),
                                      _p(idx), _p(zh), _p(bt), C.c_int32(threads))
    assert r == 0
    out = (idx,)
    if want_zhat:
        out += (zh,)
    if want_bits:
        out += (bt,)
    return out if len(out) > 1 else idx


def compress_coordinates(means, stds, beta, codebook, lengths, threads=1):
    m = np.ascontiguousarray(means, np.float32).ravel()
    s = np.ascontiguousarray(stds, np.float32).ravel()
    cb = np.ascontiguousarray(codebook, np.float64)
    ln = np.ascontiguousarray(lengths, np.int64)
    val = np.empty(m.shape, np.float32)
    slot = np.empty(m.shape, np.int32)
    r = lib().vbq_oracle_compress_coordinates(_p(m), _p(s), C.c_int64(m.shape[0]), _p(cb), _p(ln),
                                              C.c_int32(cb.shape[0]), C.c_double(float(beta)), _p(val), _p(slot),
                                              C.c_int32(threads))
    assert r == 0
    return val.reshape(np.shape(means)), slot.reshape(np.shape(means))


def histogram(idx, n_ch, N=10, layout=0):
    idx = np.ascontiguousarray(idx, np.uint16)
    L = idx.shape[0]
    E = idx[0].size
    rows = E // n_ch
    counts = np.zeros((L, n_ch, 2 ** (N + 1) - 1), np.int64)
    r = lib().vbq_oracle_histogram(_p(idx), C.c_int64(rows), C.c_int32(n_ch), C.c_int32(layout), C.c_int32(L),
                                   C.c_int32(N), _p(counts))
    assert r == 0
    return counts


def moments(x, n_ch, layout=0):
    x = np.ascontiguousarray(x, np.float32)
    rows = x.size // n_ch
    out = np.zeros((n_ch, 2), np.float64)
    r = lib().vbq_oracle_moments(_p(x), C.c_int64(rows), C.c_int32(n_ch), C.c_int32(layout), _p(out))
    assert r == 0
    return out


def numpy_sum_sq_f32(x):
    """np.sum(x.ravel()**2) for float32 x in NumPy's own (pairwise, float32) order -> np.float32."""
    x = np.ascontiguousarray(x, np.float32).ravel()
    out = C.c_float(0)
    r = lib().vbq_oracle_numpy_sum_sq_f32(_p(x), C.c_int64(x.size), C.byref(out))
    assert r == 0
    return np.float32(out.value)


def rans_encode(idx, freq, seg):
    """idx u16 [S, n], freq u16 [S, T] -> (words u16 [S, nseg, seg+2], sizes u32 [S, nseg])."""
    idx = np.ascontiguousarray(idx, np.uint16)
    freq = np.ascontiguousarray(freq, np.uint16)
    S, n = idx.shape
    nseg = (n + seg - 1) // seg
    words = np.zeros((S, nseg, seg + 2), np.uint16)
    sizes = np.zeros((S, nseg), np.uint32)
    r = lib().vbq_oracle_rans_encode(_p(idx), C.c_int64(S), C.c_int64(n), C.c_int32(freq.shape[1]), C.c_int32(seg),
                                     _p(freq), _p(words), _p(sizes))
    assert r == 0, r
    return words, sizes


def rans_decode(words, sizes, freq, n, seg):
    words = np.ascontiguousarray(words, np.uint16)
    sizes = np.ascontiguousarray(sizes, np.uint32)
    freq = np.ascontiguousarray(freq, np.uint16)
    S = freq.shape[0]
    idx = np.zeros((S, n), np.uint16)
    r = lib().vbq_oracle_rans_decode(_p(words), _p(sizes), C.c_int64(S), C.c_int64(n), C.c_int32(freq.shape[1]),
                                     C.c_int32(seg), _p(freq), _p(idx))
    assert r == 0, r
    return idx


def analogy_ranks(emb, analogies, threads=1):
    """prediction_ranks with the GPU path's documented arithmetic (fma chain in k order)."""
    e = np.ascontiguousarray(emb, np.float32)
    an = np.ascontiguousarray(analogies, np.int32)
    out = np.empty(an.shape[0], np.int64)
    r = lib().vbq_oracle_analogy_ranks(_p(e), C.c_int64(e.shape[0]), C.c_int32(e.shape[1]), _p(an), C.c_int64(an.shape[0]),
                                       _p(out), C.c_int32(threads))
    assert r == 0
    return out

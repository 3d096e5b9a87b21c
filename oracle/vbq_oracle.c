/*
 * vbq_oracle.c -- CPU restatement of the VBQ hot path in plain C.
 *
 * TEST INFRASTRUCTURE ONLY.  Loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py; never by the product package vbq_amd.
 *
 * It follows the reference formulation, not the GPU kernel's: every bit level is searched
 * separately with a lower bound on that level's own points and the result is pushed
 * through the index arithmetic of the edge-padded search grids
 *   (img-compression/quantizer.py:50-63 grids, :65-80 search + clip),
 * the 2N+1 candidates are scored in the order [L_0..L_N, R_1..R_N] (quantizer.py:183)
 * with  -0.5*((P-mu)/sigma)**2 - lambda*len  as separately rounded f32 operations
 * (img-compression/utils.py:319-320, 388-396) and the FIRST maximum wins (utils.py:401).
 * Build with -ffp-contract=off (see Makefile) so that no a*b+c is fused.
 *
 * Pinning: tests/test_oracle_c.py checks this file against the golden vectors captured
 * from the reference (tests/golden/g5, g6, g7, g8) and against oracle/vbq_oracle.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN 15

static inline int64_t elem_offset(int64_t row, int32_t c, int64_t n_rows, int32_t n_ch, int32_t layout) {
    return layout == 0 ? row * (int64_t)n_ch + c : (int64_t)c * n_rows + row;
}

/* first i in [0, m] with p[i] >= z  (np/tf searchsorted side='left') */
static inline int lower_bound_f32(const float *p, int m, float z) {
    int lo = 0, hi = m;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (p[mid] < z) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* positions (within level n) of the left/right endpoint the reference's padded grid yields */
static inline void level_endpoints(const float *lvl, int n, int N, float z, int *lpos, int *rpos) {
    const int m = 1 << n;
    if (n == 0) { *lpos = *rpos = 0; return; }                 /* quantizer.py:54-55: grid is one value */
    const int G = 1 << N;
    const int pad = (1 << (N - 1)) - (1 << (n - 1));            /* quantizer.py:57 */
    const int i = lower_bound_f32(lvl, m, z);
    int g;                                                       /* lower bound inside the padded grid */
    if (i == 0) g = 0;                                           /* z <= first point: hits the left padding */
    else if (i == m) g = G;                                      /* beyond everything */
    else g = pad + i;
    if (g > G - 1) g = G - 1;                                    /* quantizer.py:75 */
    int gl = g - 1; if (gl < 0) gl = 0;                          /* quantizer.py:76 */
    int r = g - pad;  if (r < 0) r = 0;  if (r > m - 1) r = m - 1;
    int l = gl - pad; if (l < 0) l = 0;  if (l > m - 1) l = m - 1;
    *lpos = l; *rpos = r;
}

int vbq_oracle_quantize_f32(const float *mu, const float *sigma, int64_t n_rows, int32_t n_ch, int32_t layout,
                            const float *table_lm, const float *level_len, const double *lambdas, int32_t L,
                            int32_t N, int32_t mode, uint16_t *out_idx, float *out_zhat, float *out_bits,
                            int32_t n_threads) {
    if (N < 1 || N > MAXN || n_ch < 1 || L < 1) return -1;
    const int T = (2 << N) - 1, N1 = N + 1, M = 2 * N + 1;
    const int64_t E = n_rows * (int64_t)n_ch;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < n_rows; ++row) {
        for (int32_t c = 0; c < n_ch; ++c) {
            const int64_t e = elem_offset(row, c, n_rows, n_ch, layout);
            const float z = mu[e], s = sigma[e];
            const float *tb = table_lm + (int64_t)c * T;
            float P[2 * MAXN + 1], D[2 * MAXN + 1];
            int pos[2 * MAXN + 1], lev[2 * MAXN + 1];
            for (int n = 0; n <= N; ++n) {
                int lp, rp;
                level_endpoints(tb + ((1 << n) - 1), n, N, z, &lp, &rp);
                P[n] = tb[(1 << n) - 1 + lp]; pos[n] = lp; lev[n] = n;
                if (n >= 1) { P[N + n] = tb[(1 << n) - 1 + rp]; pos[N + n] = rp; lev[N + n] = n; }
            }
            for (int j = 0; j < M; ++j) {                         /* utils.py:319-320 */
                volatile float d = P[j] - z;
                volatile float t = d / s;
                volatile float q = t * t;
                D[j] = -0.5f * q;
            }
            for (int l = 0; l < L; ++l) {
                const float *ll = level_len ? level_len + ((int64_t)l * n_ch + c) * N1 : NULL;
                int bj = 0;
                if (mode == 0) {
                    const float lam = (float)lambdas[l];
                    float best = 0.f;
                    for (int j = 0; j < M; ++j) {
                        const float len = ll ? ll[lev[j]] : (float)lev[j];
                        volatile float pen = lam * len;
                        volatile float sc = D[j] - pen;
                        if (j == 0 || sc > best) { best = sc; bj = j; }
                    }
                } else {
                    const double lam = lambdas[l];
                    double best = 0.0;
                    for (int j = 0; j < M; ++j) {
                        const double len = ll ? (double)ll[lev[j]] : (double)lev[j];
                        volatile double pen = lam * len;
                        volatile double sc = (double)D[j] - pen;
                        if (j == 0 || sc > best) { best = sc; bj = j; }
                    }
                }
                const int64_t o = (int64_t)l * E + e;
                out_idx[o] = (uint16_t)((((2 * pos[bj] + 1)) << (N - lev[bj])) - 1);
                if (out_zhat) out_zhat[o] = P[bj];
                if (out_bits) out_bits[o] = ll ? ll[lev[bj]] : (float)lev[bj];
            }
        }
    }
    return 0;
}

/* ipynb:429-443, literally: every code point of the level-major code book is scored in f64,
 * penalty (2*beta)*sigma^2 held in f32 (NumPy-1.17 casting) times the integer length,
 * first minimum wins.  out_slot = winning level-major slot. */
int vbq_oracle_compress_coordinates(const float *means, const float *stds, int64_t n, const double *codebook,
                                    const int64_t *lengths, int32_t T, double beta, float *out_val,
                                    int32_t *out_slot, int32_t n_threads) {
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    const float tb = (float)(2.0 * beta);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double z = (double)means[i];
        volatile float var = stds[i] * stds[i];
        volatile float w32 = tb * var;
        const double w = (double)w32;
        double best = 0.0; int bj = 0;
        for (int j = 0; j < T; ++j) {
            volatile double d = codebook[j] - z;
            volatile double err = d * d;
            volatile double pen = w * (double)lengths[j];
            volatile double cost = err + pen;
            if (j == 0 || cost < best) { best = cost; bj = j; }
        }
        if (out_val) out_val[i] = (float)codebook[bj];
        if (out_slot) out_slot[i] = bj;
    }
    return 0;
}

/* quantizer.py:104-105,138-140: per-(lambda, channel) bincount of rank indices */
int vbq_oracle_histogram(const uint16_t *idx, int64_t n_rows, int32_t n_ch, int32_t layout, int32_t L, int32_t N,
                         int64_t *counts) {
    const int T = (2 << N) - 1;
    const int64_t E = n_rows * (int64_t)n_ch;
    for (int l = 0; l < L; ++l)
        for (int64_t row = 0; row < n_rows; ++row)
            for (int32_t c = 0; c < n_ch; ++c) {
                const int64_t e = elem_offset(row, c, n_rows, n_ch, layout);
                counts[((int64_t)l * n_ch + c) * T + idx[(int64_t)l * E + e]] += 1;
            }
    return 0;
}

/* f64 sums of x and x^2 per channel */
int vbq_oracle_moments(const float *x, int64_t n_rows, int32_t n_ch, int32_t layout, double *out) {
    for (int32_t c = 0; c < n_ch; ++c) {
        long double s1 = 0, s2 = 0;
        for (int64_t row = 0; row < n_rows; ++row) {
            const double v = x[elem_offset(row, c, n_rows, n_ch, layout)];
            s1 += v; s2 += v * v;
        }
        out[2 * c] = (double)s1; out[2 * c + 1] = (double)s2;
    }
    return 0;
}

/* NumPy's float32 pairwise summation (numpy/core/src/umath/loops_utils.h.src, FLOAT_pairwise_sum: unrolled blocks of
 * at most 128 elements with 8 accumulators, recursive halving on multiples of 8 above that) applied to x[i]*x[i] with
 * the product rounded to float32 first, in blocks of 8192: np.sum(x.ravel()**2) for a contiguous float32 x -- the reduction behind the
 * notebook's empirical_std = np.sqrt(np.mean(vecs_u.ravel()**2)) (ipynb:374).  Pinned against NumPy itself in
 * tests/test_oracle_c.py (NumPy is what the reference runs). */
static float pw_sq(const float *a, int64_t n) {
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; ++i) res += a[i] * a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j] * a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j] * a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i] * a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pw_sq(a, n2) + pw_sq(a + n2, n - n2);
}

/* The reduction loop hands the pairwise routine at most 8192 elements at a time (NumPy's iterator buffer size) and
 * adds the block results one after the other: acc = fl(acc + pairwise(block)). */
int vbq_oracle_numpy_sum_sq_f32(const float *x, int64_t n, float *out) {
    float acc = 0.0f;
    for (int64_t i = 0; i < n; i += 8192) acc += pw_sq(x + i, n - i < 8192 ? n - i : 8192);
    *out = acc;
    return 0;
}

int vbq_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * rANS segment coder (checker for vbq_amd/csrc/vbq_rans.hip; format documented in include/vbq.h).
 * 32-bit state, 16-bit renormalisation, 15 probability bits; per segment: emission-order words,
 * then the final state (low, high).  Textbook rANS (Duda 2013; byte-wise form after F. Giesen's
 * public-domain ryg_rans), restated for 16-bit words.
 * ------------------------------------------------------------------------------------------ */
#define RANS_PB 15
#define RANS_L (1u << 16)

int vbq_oracle_rans_encode(const uint16_t *idx, int64_t n_streams, int64_t n, int32_t T, int32_t seg,
                           const uint16_t *freq, uint16_t *words, uint32_t *sizes) {
    const int64_t nseg = (n + seg - 1) / seg;
    uint32_t *cum = (uint32_t *)malloc(sizeof(uint32_t) * (T + 1));
    for (int64_t s = 0; s < n_streams; ++s) {
        const uint16_t *f = freq + s * T;
        cum[0] = 0;
        for (int i = 0; i < T; ++i) cum[i + 1] = cum[i] + f[i];
        if (cum[T] != (1u << RANS_PB)) { free(cum); return -2; }
        for (int64_t g = 0; g < nseg; ++g) {
            const int64_t a = g * seg, b = (a + seg < n) ? a + seg : n;
            uint16_t *out = words + (s * nseg + g) * (int64_t)(seg + 2);
            uint32_t x = RANS_L;
            int k = 0;
            for (int64_t i = b - 1; i >= a; --i) {
                const uint32_t sym = idx[s * n + i];
                const uint32_t fs = f[sym];
                if (x >= (fs << (32 - RANS_PB))) { out[k++] = (uint16_t)(x & 0xffffu); x >>= 16; }
                x = ((x / fs) << RANS_PB) + (x % fs) + cum[sym];
            }
            out[k++] = (uint16_t)(x & 0xffffu);
            out[k++] = (uint16_t)(x >> 16);
            sizes[s * nseg + g] = (uint32_t)k;
        }
    }
    free(cum);
    return 0;
}

int vbq_oracle_rans_decode(const uint16_t *words, const uint32_t *sizes, int64_t n_streams, int64_t n, int32_t T,
                           int32_t seg, const uint16_t *freq, uint16_t *idx) {
    const int64_t nseg = (n + seg - 1) / seg;
    uint32_t *cum = (uint32_t *)malloc(sizeof(uint32_t) * (T + 1));
    uint16_t *lut = (uint16_t *)malloc(sizeof(uint16_t) << RANS_PB);
    for (int64_t s = 0; s < n_streams; ++s) {
        const uint16_t *f = freq + s * T;
        cum[0] = 0;
        for (int i = 0; i < T; ++i) cum[i + 1] = cum[i] + f[i];
        if (cum[T] != (1u << RANS_PB)) { free(cum); free(lut); return -2; }
        for (int i = 0; i < T; ++i)
            for (uint32_t j = cum[i]; j < cum[i + 1]; ++j) lut[j] = (uint16_t)i;
        for (int64_t g = 0; g < nseg; ++g) {
            const int64_t a = g * seg, b = (a + seg < n) ? a + seg : n;
            const uint16_t *in = words + (s * nseg + g) * (int64_t)(seg + 2);
            int k = (int)sizes[s * nseg + g];
            uint32_t x = ((uint32_t)in[k - 1] << 16) | in[k - 2];
            k -= 2;
            for (int64_t i = a; i < b; ++i) {
                const uint32_t slot = x & ((1u << RANS_PB) - 1u);
                const uint32_t sym = lut[slot];
                idx[s * n + i] = (uint16_t)sym;
                x = f[sym] * (x >> RANS_PB) + slot - cum[sym];
                if (x < RANS_L) x = (x << 16) | in[--k];
            }
            if (k != 0 || x != RANS_L) { free(cum); free(lut); return -3; }   /* stream must be consumed exactly */
        }
    }
    free(cum);
    free(lut);
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * f4: prediction_ranks (compress-trained-word-embeddings.ipynb cell 14, ipynb:199-209), restated with
 * the arithmetic the GPU path documents: f32 row norms summed in index order, IEEE division,
 * (normed[b] - normed[a]) + normed[c], scores as an fma chain over ascending k, strict '<'.
 * NumPy/BLAS (the reference) rounds the dot products in a different order; tests compare with it
 * through a near-tie tolerance and with this function bit for bit.
 * --------------------------------------------------------------------------------------------- */
int vbq_oracle_analogy_ranks(const float *emb, int64_t V, int32_t K, const int32_t *an, int64_t Q, int64_t *ranks,
                             int32_t threads) {
    float *nrm = (float *)malloc(sizeof(float) * (size_t)V * K);
    if (!nrm) return -1;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (int64_t v = 0; v < V; ++v) {
        float s = 0.0f;
        for (int k = 0; k < K; ++k) s = s + emb[v * K + k] * emb[v * K + k];
        const float den = 1e-8f + sqrtf(s);
        for (int k = 0; k < K; ++k) nrm[v * K + k] = emb[v * K + k] / den;
    }
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < Q; ++i) {
        const float *a = nrm + (int64_t)an[4 * i] * K, *b = nrm + (int64_t)an[4 * i + 1] * K;
        const float *c = nrm + (int64_t)an[4 * i + 2] * K, *d = nrm + (int64_t)an[4 * i + 3] * K;
        float *p = (float *)malloc(sizeof(float) * (size_t)K);
        float gt = 0.0f;
        for (int k = 0; k < K; ++k) {
            p[k] = (b[k] - a[k]) + c[k];
            gt = fmaf(p[k], d[k], gt);
        }
        int64_t below = 0;
        for (int64_t v = 0; v < V; ++v) {
            float s = 0.0f;
            for (int k = 0; k < K; ++k) s = fmaf(p[k], nrm[v * K + k], s);
            below += s < gt;
        }
        ranks[i] = V - below - 1;
        free(p);
    }
    free(nrm);
    return 0;
}

// K1n: the word-embedding variant of the solve (compress_coordinates,
// word-embeddings/compress-trained-word-embeddings.ipynb:429-443) in its own arithmetic:
//   squared error  fl64(fl64(c - f64(mu))^2)           (ipynb:436, f64 code book minus f32 means)
//   penalty        f64(fl32(fl32(2 beta) * fl32(sigma^2))) * len   (ipynb:438 under NumPy-1.17 casting)
//   argmin over the code book in level-major order, first minimum wins (ipynb:440)
// The notebook scores all 2047 points; only the two neighbours of mu on each bit level can
// win (rounding is monotone, so a farther point of the same level never scores lower), so
// the same 21-candidate descent as K1 is used, scanned in level-major order.
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "vbq_common.h"

namespace vbq {
namespace {

constexpr int kMaxBetaChunk = 64;
struct BetaChunk {
    double beta[kMaxBetaChunk];
};

template <int N>
struct ElemD {
    double a[N + 1];
    double b[N + 1];
    uint32_t G;
};

__device__ __forceinline__ double sq_err(double c, double z) {
    const double d = __dsub_rn(c, z);
    return __dmul_rn(d, d);
}

// v_min_f64 as it is (fmin makes the compiler canonicalise both operands: two v_max_f64 x, x per call)
__device__ __forceinline__ double nb_min64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int N>
__device__ __forceinline__ void search_and_score_d(const double *tb, double z, ElemD<N> &e) {
    uint32_t g = 0;
#pragma unroll
    for (int n = 0; n <= N; ++n) {
        const int off = (1 << n) - 1;
        const int m = 1 << n;
        const uint32_t j = g;
        const double pj = tb[off + j];
        const bool below = pj < z;
        if (n == 0) {
            e.a[0] = e.b[0] = sq_err(pj, z);
        } else {
            int jo = below ? (int)j + 1 : (int)j - 1;
            jo = jo < 0 ? 0 : (jo > m - 1 ? m - 1 : jo);
            const double po = tb[off + jo];
            const double sj = sq_err(pj, z), so = sq_err(po, z);
            e.a[n] = below ? sj : so;
            e.b[n] = below ? so : sj;
        }
        g = 2 * g + (below ? 1u : 0u);
    }
    e.G = g;
}

template <int N>
__global__ void __launch_bounds__(256)
k_quant_notebook(const float *__restrict__ means, const float *__restrict__ stds, long n,
                 const double *__restrict__ codebook, BetaChunk bc, int nb,
                 uint16_t *__restrict__ out_idx, float *__restrict__ out_val, int vec_ok) {
    constexpr int T = table_size(N);
    __shared__ double tb[T + 1];
    for (int i = threadIdx.x; i < T; i += blockDim.x) tb[i] = codebook[i];
    __syncthreads();
    const long npairs = (n + 1) >> 1;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < npairs; q += (long)gridDim.x * blockDim.x) {
        const long i0 = q * 2;
        const bool full = vec_ok && (i0 + 2 <= n);
        float m2[2], s2[2];
        if (full) {
            const float2 mv = *reinterpret_cast<const float2 *>(means + i0);
            const float2 sv = *reinterpret_cast<const float2 *>(stds + i0);
            m2[0] = mv.x; m2[1] = mv.y; s2[0] = sv.x; s2[1] = sv.y;
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bool ok = i0 + k < n;
                m2[k] = ok ? means[i0 + k] : 0.0f;
                s2[k] = ok ? stds[i0 + k] : 1.0f;
            }
        }
        ElemD<N> el[2];
        float var[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            search_and_score_d<N>(tb, (double)m2[k], el[k]);
            var[k] = __fmul_rn(s2[k], s2[k]);
        }
        for (int l = 0; l < nb; ++l) {
            const float tb2 = (float)(2.0 * bc.beta[l]);
            uint32_t idx[2];
            float val[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double w = (double)__fmul_rn(tb2, var[k]);
                double best = el[k].a[0];     // length 0: no penalty (w * 0 == 0, x + 0 == x)
                int lvl = 0;
                bool right = false;
#pragma unroll
                for (int nn = 1; nn <= N; ++nn) {
                    const double pen = __dmul_rn(w, (double)nn);
                    const double ca = __dadd_rn(el[k].a[nn], pen);
                    bool up = ca < best;
                    best = up ? ca : best; lvl = up ? nn : lvl; right = up ? false : right;
                    const double cb = __dadd_rn(el[k].b[nn], pen);
                    up = cb < best;
                    best = up ? cb : best; lvl = up ? nn : lvl; right = up ? true : right;
                }
                const uint32_t Gn = el[k].G >> (N - lvl);
                const uint32_t g = (Gn + 1) >> 1;
                const uint32_t m = 1u << lvl;
                const uint32_t r = g < m - 1 ? g : m - 1;
                const uint32_t lft = g > 0 ? g - 1 : 0;
                const uint32_t pos = right ? r : lft;
                idx[k] = ((2 * pos + 1) << (N - lvl)) - 1;
                if (out_val) val[k] = (float)tb[(1 << lvl) - 1 + pos];
            }
            const long o = (long)l * n + i0;
            if (full) {
                *reinterpret_cast<uint32_t *>(out_idx + o) = idx[0] | (idx[1] << 16);
                if (out_val) *reinterpret_cast<float2 *>(out_val + o) = make_float2(val[0], val[1]);
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (i0 + k < n) {
                        out_idx[o + k] = (uint16_t)idx[k];
                        if (out_val) out_val[o + k] = val[k];
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fast form (same restructuring as vbq_quantize_fast.hip, in f64): per level keep the better
// side's squared error du_n and a packed word (rank of the better side + a 10-bit lower bound
// of errL - errR when the better side is R); per beta: cost_n = fma(w, n, du_n) -- the product
// w*n is exact (24-bit w, n <= 15), so one fused rounding equals fl64(err + fl64(w*n)) of
// ipynb:436-440 -- a min tree, the sign mask of S - cost_n, and the shallowest matching level.
// Level-major order means a shallower level always precedes a deeper one, so cross-level ties
// need no special care; only a right side that beats its left side by less than ~ulp(S) is
// re-solved with the literal scan.
// ------------------------------------------------------------------------------------------
template <int N>
__device__ __noinline__ uint32_t exact_rank_scan_nb(const double *tb, double z, double w) {
    uint32_t g = 0, best_rank = 0;
    double best = 0.0;
    for (int n = 0; n <= N; ++n) {
        const int off = (1 << n) - 1, m = 1 << n;
        const uint32_t j = g;
        const double pj = tb[off + j];
        const bool below = pj < z;
        const double pen = __dmul_rn(w, (double)n);
        if (n == 0) {
            best = __dadd_rn(sq_err(pj, z), pen);
            best_rank = (1u << N) - 1;
        } else {
            int jo = below ? (int)j + 1 : (int)j - 1;
            jo = jo < 0 ? 0 : (jo > m - 1 ? m - 1 : jo);
            const double po = tb[off + jo];
            const uint32_t posL = below ? j : (uint32_t)jo, posR = below ? (uint32_t)jo : j;
            const double cL = __dadd_rn(sq_err(below ? pj : po, z), pen);
            const double cR = __dadd_rn(sq_err(below ? po : pj, z), pen);
            if (cL < best) { best = cL; best_rank = ((2 * posL + 1) << (N - n)) - 1; }
            if (cR < best) { best = cR; best_rank = ((2 * posR + 1) << (N - n)) - 1; }
        }
        g = 2 * g + (below ? 1u : 0u);
    }
    return best_rank;
}

template <int M>
__device__ __forceinline__ double min_of_d(const double (&v)[M]) {
    double t[M];
#pragma unroll
    for (int i = 0; i < M; ++i) t[i] = v[i];
    int m = M;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        if (m <= 1) break;
        int o = 0;
#pragma unroll
        for (int i = 0; i + 1 < m; i += 2) t[o++] = fmin(t[i], t[i + 1]);
        if (m & 1) t[o++] = t[m - 1];
        m = o;
    }
    return t[0];
}

template <int N>
__global__ void __launch_bounds__(256)
k_quant_notebook_fast(const float *__restrict__ means, const float *__restrict__ stds, long n,
                      const double *__restrict__ codebook, BetaChunk bc, int nb,
                      uint16_t *__restrict__ out_idx, float *__restrict__ out_val, int vec_ok, int dbg) {
    constexpr int T = table_size(N);
    constexpr int N1 = N + 1;
    constexpr int NE = 2;
    __shared__ double tb[T + 1];
    __shared__ uint32_t scratch[N1 * NE * 256];
    for (int i = threadIdx.x; i < T; i += blockDim.x) tb[i] = codebook[i];
    __syncthreads();
    const long npairs = (n + 1) >> 1;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < npairs; q += (long)gridDim.x * blockDim.x) {
        const long i0 = q * 2;
        const bool full = vec_ok && (i0 + 2 <= n);
        float m2[NE], s2[NE];
        if (full) {
            const float2 mv = *reinterpret_cast<const float2 *>(means + i0);
            const float2 sv = *reinterpret_cast<const float2 *>(stds + i0);
            m2[0] = mv.x; m2[1] = mv.y; s2[0] = sv.x; s2[1] = sv.y;
        } else {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const bool ok = i0 + k < n;
                m2[k] = ok ? means[i0 + k] : 0.0f;
                s2[k] = ok ? stds[i0 + k] : 1.0f;
            }
        }
        double du[NE][N1];
        uint32_t g[NE] = {0, 0};
        float var[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) var[k] = __fmul_rn(s2[k], s2[k]);
#pragma unroll
        for (int lv = 0; lv <= N; ++lv) {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const double z = (double)m2[k];
                const int off = (1 << lv) - 1, m = 1 << lv;
                const uint32_t j = g[k];
                const double pj = tb[off + j];
                const bool below = pj < z;
                double errL, errR;
                uint32_t posL, posR;
                if (lv == 0) {
                    errL = errR = sq_err(pj, z);
                    posL = posR = 0;
                } else {
                    int jo = below ? (int)j + 1 : (int)j - 1;
                    jo = jo < 0 ? 0 : (jo > m - 1 ? m - 1 : jo);
                    const double po = tb[off + jo];
                    const double ej = sq_err(pj, z), eo = sq_err(po, z);
                    errL = below ? ej : eo; errR = below ? eo : ej;
                    posL = below ? j : (uint32_t)jo; posR = below ? (uint32_t)jo : j;
                }
                g[k] = 2 * j + (below ? 1u : 0u);
                const bool r_better = errR < errL;
                du[k][lv] = r_better ? errR : errL;
                const uint32_t rank = ((2 * (r_better ? posR : posL) + 1) << (N - lv)) - 1;
                const uint32_t gb = __float_as_uint((float)__dsub_rn(errL, errR)) >> 21;   // f32 image of the gap
                const uint32_t gcode = r_better ? (gb > 1 ? gb - 2 : 0) : 0x3ffu;           // two codes down: safely below
                scratch[(lv * NE + k) * 256 + threadIdx.x] = (gcode << 21) | rank;
            }
        }
        for (int l = 0; l < nb; ++l) {
            const float tb2 = (float)(2.0 * bc.beta[l]);
            double cst[NE][N1], S[NE], w[NE];
#pragma unroll
            for (int k = 0; k < NE; ++k) w[k] = (double)__fmul_rn(tb2, var[k]);
#pragma unroll
            for (int lv = 0; lv < N1; ++lv)
#pragma unroll
                for (int k = 0; k < NE; ++k) cst[k][lv] = lv == 0 ? du[k][0] : __fma_rn(w[k], (double)lv, du[k][lv]);
#pragma unroll
            for (int k = 0; k < NE; ++k) S[k] = min_of_d<N1>(cst[k]);
            uint32_t ne[NE] = {0, 0};
#pragma unroll
            for (int lv = N; lv >= 0; --lv)
#pragma unroll
                for (int k = 0; k < NE; ++k)
                    ne[k] = __builtin_amdgcn_alignbit(ne[k], (uint32_t)__double2hiint(__dsub_rn(S[k], cst[k][lv])), 31);
            uint32_t rank[NE];
            bool flagged[NE];
            bool any_flag = false;
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int n1 = __builtin_ctz(~ne[k]);
                const uint32_t pk = scratch[(n1 * NE + k) * 256 + threadIdx.x];
                // errL - errR < ~ulp64(S) = 2^-52 S could let fl64(errL + pen) round onto fl64(errR + pen): flag below 2^-46 S
                const uint32_t thr = __float_as_uint(__fmul_rn((float)S[k], 1.4210854715202004e-14f));
                flagged[k] = (((pk & 0x7fe00000u) <= thr) && dbg != 2) || dbg == 1;
                rank[k] = pk & 0x7ffu;
                any_flag = any_flag || flagged[k];
            }
            if (__any(any_flag)) {
#pragma unroll
                for (int k = 0; k < NE; ++k)
                    if (flagged[k]) rank[k] = exact_rank_scan_nb<N>(tb, (double)m2[k], w[k]);
            }
            float val[NE];
            if (out_val) {
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    const uint32_t kk = rank[k] + 1;
                    const int tz = __builtin_ctz(kk);
                    val[k] = (float)tb[(1 << (N - tz)) - 1 + (kk >> (tz + 1))];
                }
            }
            const long o = (long)l * n + i0;
            if (full) {
                *reinterpret_cast<uint32_t *>(out_idx + o) = rank[0] | (rank[1] << 16);
                if (out_val) *reinterpret_cast<float2 *>(out_val + o) = make_float2(val[0], val[1]);
            } else {
#pragma unroll
                for (int k = 0; k < NE; ++k)
                    if (i0 + k < n) {
                        out_idx[o + k] = (uint16_t)rank[k];
                        if (out_val) out_val[o + k] = val[k];
                    }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// K1np: compress_coordinates for ONE or TWO betas per call -- how the notebook itself calls it (test_beta, ipynb:466, in a
// loop over 50 betas, ipynb:1103) -- with exact pruning of the descent.
//
// Per bit level only the nearer neighbour's squared error is formed (monotone rounding: it is the smaller one) and
// cost_n = fl64(err_n + w n) (the product is exact: w is an f32 number).  The notebook takes the FIRST minimum in level-major
// order, so a deeper level wins only with a strictly smaller cost: the running minimum with a strict compare IS the scan
// across levels, no tie rule is left over.  Every cost is at least its penalty w n, so the descent stops for the whole wave
// once best <= w (n + 1) holds in all its lanes.  The two sides of the one winning level are compared at the end exactly as
// the scan compares them (L first; R only if strictly smaller).  Only comparisons the reference makes: no certificate, no
// fall-back.
template <int N, int LL, bool WITH_VAL>
__global__ void __launch_bounds__(256)
k_quant_notebook_pruned(const float *__restrict__ means, const float *__restrict__ stds, long n,
                        const double *__restrict__ codebook, BetaChunk bc, uint16_t *__restrict__ out_idx,
                        float *__restrict__ out_val, int vec_ok) {
    constexpr int T = table_size(N);
    constexpr int NE = 2;
    __shared__ double tb[T + 1];
    for (int i = threadIdx.x; i < T; i += blockDim.x) tb[i] = codebook[i];
    __syncthreads();
    float tb2[LL];
#pragma unroll
    for (int l = 0; l < LL; ++l) tb2[l] = (float)(2.0 * bc.beta[l]);
    const char *tbb = reinterpret_cast<const char *>(tb);
    const long npairs = (n + 1) >> 1;
    unsigned int rot = wave_slot();
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < npairs; q += (long)gridDim.x * blockDim.x) {
        rotate_issue_priority(rot);                            // resident grid: see vbq_common.h
        const long i0 = q * 2;
        const bool full = vec_ok && (i0 + 2 <= n);
        float m2[NE], s2[NE];
        if (full) {
            const float2 mv = *reinterpret_cast<const float2 *>(means + i0);
            const float2 sv = *reinterpret_cast<const float2 *>(stds + i0);
            m2[0] = mv.x; m2[1] = mv.y; s2[0] = sv.x; s2[1] = sv.y;
        } else {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const bool ok = i0 + k < n;
                m2[k] = ok ? means[i0 + k] : 0.0f;
                s2[k] = ok ? stds[i0 + k] : 1.0f;
            }
        }
        double z[NE], w[NE][LL], best[NE][LL];
        uint32_t g[NE], win[NE][LL];                           // byte slot of the visited point (bits 0 .. 13) | level << 16
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            z[k] = (double)m2[k];
            g[k] = 0;
            const float var = __fmul_rn(s2[k], s2[k]);
#pragma unroll
            for (int l = 0; l < LL; ++l) {
                w[k][l] = (double)__fmul_rn(tb2[l], var);      // fl32(fl32(2 beta) * fl32(sigma^2)), ipynb:438 under NumPy 1.17
                best[k][l] = 0.0;
                win[k][l] = 0;
            }
        }
        bool done = false;                                     // wave-uniform; a flag instead of a break keeps the loop unrollable
#pragma unroll
        for (int lv = 0; lv <= N; ++lv) {
            if (done) continue;
            const int off8 = 8 * ((1 << lv) - 1);
            const int top8 = off8;
            uint64_t active = 0;
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const double pj = *reinterpret_cast<const double *>(tbb + off8 + g[k]);
                const double d1 = __dsub_rn(pj, z[k]);
                const bool below = pj < z[k];
                double err;
                if (lv == 0) {
                    err = __dmul_rn(d1, d1);
                } else {
                    int o8 = (int)g[k] + (below ? 8 : -8);     // the other neighbour sits on z's side of the visited point
                    o8 = o8 < 0 ? 0 : (o8 > top8 ? top8 : o8);
                    const double po = *reinterpret_cast<const double *>(tbb + off8 + o8);
                    const double d2 = __dsub_rn(po, z[k]);
                    const double dn = nb_min64(fabs(d1), fabs(d2));
                    err = __dmul_rn(dn, dn);
                }
                const uint32_t here = g[k] | ((uint32_t)lv << 16);
#pragma unroll
                for (int l = 0; l < LL; ++l) {
                    const double cst = lv == 0 ? err : __fma_rn(w[k][l], (double)lv, err);
                    const bool lt = lv == 0 || cst < best[k][l];
                    best[k][l] = lt ? cst : best[k][l];
                    win[k][l] = lt ? here : win[k][l];
                    if (lv < N) active |= __builtin_amdgcn_ballot_w64(!(best[k][l] <= __dmul_rn(w[k][l], (double)(lv + 1))));
                }
                g[k] = 2 * g[k] + (below ? 8u : 0u);
            }
            if (lv < N && active == 0) done = true;            // no deeper level can cost less in this wave
        }
        uint32_t rank[LL][NE];
        float val[LL][NE];
#pragma unroll
        for (int k = 0; k < NE; ++k)
#pragma unroll
            for (int l = 0; l < LL; ++l) {
                const uint32_t lv = win[k][l] >> 16;
                const uint32_t gj = win[k][l] & 0xffffu;
                const uint32_t off8 = 8u * ((1u << lv) - 1u), top8 = off8;
                const double pj = *reinterpret_cast<const double *>(tbb + off8 + gj);
                const bool below = pj < z[k];
                int o8 = (int)gj + (below ? 8 : -8);
                o8 = o8 < 0 ? 0 : (o8 > (int)top8 ? (int)top8 : o8);
                const double po = *reinterpret_cast<const double *>(tbb + off8 + o8);
                const double pen = __dmul_rn(w[k][l], (double)lv);
                const double cj = __dadd_rn(sq_err(pj, z[k]), pen), co = __dadd_rn(sq_err(po, z[k]), pen);
                // L is the visited point when it lies below z; the scan takes L first and R only if strictly smaller
                const bool take_j = below ? !(co < cj) : (cj < co);
                const uint32_t b8 = (lv == 0u || take_j) ? gj : (uint32_t)o8;
                rank[l][k] = (((b8 >> 2) + 1u) << ((uint32_t)N - lv)) - 1u;          // ((2 pos + 1) << (N - lv)) - 1, pos = b8 / 8
                if (WITH_VAL) val[l][k] = (float)((lv == 0u || take_j) ? pj : po);
            }
#pragma unroll
        for (int l = 0; l < LL; ++l) {
            const long o = (long)l * n + i0;
            if (full) {
                *reinterpret_cast<uint32_t *>(out_idx + o) = rank[l][0] | (rank[l][1] << 16);
                if (WITH_VAL) *reinterpret_cast<float2 *>(out_val + o) = make_float2(val[l][0], val[l][1]);
            } else {
#pragma unroll
                for (int k = 0; k < NE; ++k)
                    if (i0 + k < n) {
                        out_idx[o + k] = (uint16_t)rank[l][k];
                        if (WITH_VAL) out_val[o + k] = val[l][k];
                    }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// K1nt: the beta sweep without a per-beta argmin (N = 10).
//
// The notebook's cost of bit level n is a LINE in the penalty weight, err_n + w * n with w = fl32(fl32(2 beta) * fl32(sigma^2))
// (ipynb:438-440), and the same eleven lines serve every beta of the sweep.  As in K1t (vbq_quantize_fast.hip) the winner is the
// lower envelope of the lines, described by ten thresholds per element,
//       T_n = max_{j > n} min_{i <= n} (err_i - err_j) / (j - i),          T_0 >= T_1 >= ... >= T_9,
// and the level at w is #{ n : w < T_n }.  Here the kernel must produce an INDEX per (element, beta), so the thresholds are
// turned into positions a_n = #{ l : b_(l) < T_n / sigma^2 } in the SORTED sweep b_(l) = fl32(2 beta_l) (bucket table + one
// compare, as K1t), the positions into a per-lane column "levels lost at sweep point l" (ten LDS adds), and the sweep is
// then emitted by walking that column: per solve one 4-bit field, one add and one LDS read of the rank the element has on
// its current level -- about 3 vector instructions per solve instead of ~50.
//
// Exactness.  The reference compares fl64(err_n + w n) (w n is exact) and takes the first minimum in level-major order, i.e.
// the shallowest level among equal costs and, inside a level, the left neighbour unless the right one is strictly better.
// Away from every threshold the envelope's margin over any other line is at least |w - T| (integer slopes), so the rounded
// comparison agrees with the real one when that exceeds 2^-51 of the envelope's height.  Thresholds are computed in f32 from
// dv_n = f32(err_n) / sigma^2 (every dv_n within 2^-23 of its value, relatively, plus a common factor): a threshold then
// carries at most 2^-21.4 |T| + 2^-21 dv_n of error (a level j whose dv_j exceeds 2 dv_n only produces negative candidates
// and cannot raise the maximum above zero), and fl32(b var) is within 2^-24 of b var.  Every threshold therefore gets K1t's
// guard band |b - T_n / var| <= 2^-20 (dv_n + |T_n / var| (n + 1)); sweep points inside a band are re-solved for that
// element with the literal scan (exact_rank_scan_nb) and overwritten.  The whole element takes the literal scan when sigma^2
// or a distortion leaves the range in which those bounds hold, or when on some level the right neighbour beats the left one by
// less than 2^-45 of err_0, the largest cost a winner can have (fl64(errL + pen) could round onto fl64(errR + pen) and hand
// the win to the left point).
// ------------------------------------------------------------------------------------------
constexpr int kNbKeys = 1536;            // 24 octaves of 64 buckets (the notebook's sweep spans 23.3)
constexpr int kNbKeyShift = 17;          // key = float bits >> 17: sign, exponent, 6 mantissa bits
struct NbSweep {
    float b[kMaxBetaChunk];              // fl32(2 beta), ascending; +big beyond L
    unsigned char perm[kMaxBetaChunk];   // position of b[l] in the caller's order
    int L, key0, nkeys;                  // keys = float bits >> kNbKeyShift; bucket k <-> key0 + k
    float var_lo, var_hi;                // sigma^2 range in which every fl32(b var) is a normal number
    unsigned char lut[kNbKeys];          // lut[k] = #{ l : b[l] below the lower edge of bucket k }
};
constexpr float kNbBig = 3.0e38f;

__device__ __forceinline__ float nb_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float nb_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float nb_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float nb_min3abs(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// The ten thresholds T_n = max_{j > n} min_{i <= n} (dv_i - dv_j) / (j - i) of an element (see the kernel's header).
__device__ __forceinline__ void nb_thresholds(const float (&dv)[11], float (&Tn)[10]) {
    constexpr int N = 10, N1 = 11;
    float Pm[N1];
#pragma unroll
    for (int nn = 0; nn < N; ++nn) {
#pragma unroll
        for (int j = nn + 1; j < N1; ++j) {
            const float r = __fmul_rn(__fsub_rn(dv[nn], dv[j]), 1.0f / (float)(j - nn));
            Pm[j] = nn == 0 ? r : nb_min(Pm[j], r);
        }
        float t = Pm[nn + 1];
        int j = nn + 2;
#pragma unroll
        for (; j + 1 < N1; j += 2) t = nb_max3(t, Pm[j], Pm[j + 1]);
        if (j < N1) t = nb_max(t, Pm[j]);
        Tn[nn] = nb_min(t, 1.0e38f);
    }
}

// CW: words of eight 4-bit fields per column (positions 0 .. L: 5 for L <= 32, 9 for L <= 64).  NF > 0: the sweep has exactly NF full
// words (L >> 3 == NF) and they are emitted as ONE straight block -- all 16 NF rank reads of a lane pair in flight together --
// instead of NF blocks with a branch and a full LDS latency each; a last partial word takes the generic code.
// BUF: all index rows lie within 4 GB of out_idx (the host checks): the straight block's stores go through one buffer descriptor
// with the row's byte offset as the instruction's scalar offset (no 64-bit address arithmetic per store).
template <bool WITH_VAL, int CW, int NF, bool BUF>
__global__ void __launch_bounds__(256, WITH_VAL ? 2 : (CW <= 5 ? 4 : 3))
k_quant_notebook_hull(const float *__restrict__ means, const float *__restrict__ stds, long n,
                      const double *__restrict__ codebook, NbSweep sw, uint16_t *__restrict__ out_idx,
                      float *__restrict__ out_val, int vec_ok, int dbg) {
    constexpr int N = 10, N1 = 11, NE = 2;
    constexpr int T = table_size(N);
    // One LDS block.  The byte offset of rk[r][k][tid] inside the rank table is (r << 10) | (k * 512 + tid * 2): the lane part
    // stays below 1024, so one v_and_or_b32 forms the address from a shifted 4-bit field (emission below).
    struct Lds {
        unsigned short rk[N1 * NE * 256];                     // rank of the better side, [N - level][element][thread]
        unsigned char lut[kNbKeys];
        float4 rec[kMaxBetaChunk + 2];                        // rec[i] = { b[i-1], b[i], b[i+1], - } with -big / +big outside
        unsigned char perm_s[kMaxBetaChunk];
        double tb[T + 1];
        unsigned int cnt[CW * NE * 256];                      // [word][element][thread]: levels lost at sweep point l
        float vl[WITH_VAL ? N1 * NE * 256 : 1];               // the better side's code point, [N - level][element][thread]
    };
    __shared__ __align__(16) Lds lds;
    unsigned short *rk = lds.rk;
    unsigned char *lut = lds.lut;
    float4 *rec = lds.rec;
    unsigned char *perm_s = lds.perm_s;
    double *tb = lds.tb;
    unsigned int *cnt = lds.cnt;
    float *vl = lds.vl;
    const int L = sw.L;
    const unsigned int tid = threadIdx.x;
    for (int i = tid; i < T; i += blockDim.x) tb[i] = codebook[i];
    for (int k = tid; k < kNbKeys / 4; k += blockDim.x)       // the bucket table travels in the kernel arguments
        reinterpret_cast<uint32_t *>(lut)[k] = reinterpret_cast<const uint32_t *>(sw.lut)[k];
    if (tid < kMaxBetaChunk + 2) {
        const int i = (int)tid;
        auto at = [&](int l) { return l < 0 ? -kNbBig : (l < L ? sw.b[l < kMaxBetaChunk ? l : kMaxBetaChunk - 1] : kNbBig); };
        rec[i] = make_float4(at(i - 1), at(i), at(i + 1), 0.0f);
    }
    for (int i = tid; i < CW * NE * 256; i += blockDim.x) cnt[i] = 0;
    if (tid < kMaxBetaChunk) perm_s[tid] = sw.perm[tid];
    __syncthreads();

    const bool force_slow = dbg == 1, never_flag = dbg == 2;
    const uint64_t all_l = L >= 64 ? ~0ull : ((1ull << L) - 1ull);
    const unsigned int lane = tid & 63u;
    const long npairs = (n + 1) >> 1;
    const int key0 = sw.key0, nkeys = sw.nkeys;
    unsigned int rot = wave_slot();
    for (long q = (long)blockIdx.x * blockDim.x + tid; q < npairs; q += (long)gridDim.x * blockDim.x) {
        rotate_issue_priority(rot);                            // resident grid: see vbq_common.h
        const long i0 = q * 2;
        const bool full = vec_ok && (i0 + 2 <= n);
        float m2[NE], s2[NE];
        if (full) {
            const float2 mv = *reinterpret_cast<const float2 *>(means + i0);
            const float2 sv = *reinterpret_cast<const float2 *>(stds + i0);
            m2[0] = mv.x; m2[1] = mv.y; s2[0] = sv.x; s2[1] = sv.y;
        } else {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const bool ok = i0 + k < n;
                m2[k] = ok ? means[i0 + k] : 0.0f;
                s2[k] = ok ? stds[i0 + k] : 1.0f;
            }
        }
        // ---- per level: both neighbours' squared errors as the notebook forms them, the better side's rank (and point)
        float dv[NE][N1], var[NE], rv[NE];
        bool slow[NE];
        {
            uint32_t g[NE] = {0, 0};
            uint32_t hi_d0[NE] = {0, 0};
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                var[k] = __fmul_rn(s2[k], s2[k]);
                rv[k] = __builtin_amdgcn_rcpf(var[k]);
                slow[k] = !(var[k] >= sw.var_lo && var[k] <= sw.var_hi) || force_slow;
            }
#pragma unroll
            for (int lv = 0; lv <= N; ++lv) {
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    const double z = (double)m2[k];
                    const int off = (1 << lv) - 1, m = 1 << lv;
                    const uint32_t j = g[k];
                    const double pj = tb[off + j];
                    const bool below = pj < z;
                    double du;
                    uint32_t pos;
                    double pt = pj;
                    if (lv == 0) {
                        du = sq_err(pj, z);
                        pos = 0;
                        hi_d0[k] = (uint32_t)__double2hiint(du);
                    } else {
                        int jo = below ? (int)j + 1 : (int)j - 1;
                        jo = jo < 0 ? 0 : (jo > m - 1 ? m - 1 : jo);
                        const double po = tb[off + jo];
                        const double ej = sq_err(pj, z), eo = sq_err(po, z);
                        // errL = below ? ej : eo, errR = below ? eo : ej (the visited point is L when it lies below z), without the
                        // four selects: the better side's error is the minimum, and "R strictly better" is a sign of ej - eo
                        const double diff = __dsub_rn(ej, eo);             // = errL - errR when below, errR - errL otherwise (exact negation)
                        const bool r_better = below ? diff > 0.0 : diff < 0.0;
                        du = nb_min64(ej, eo);
                        pos = (r_better != below) ? j : (uint32_t)jo;      // L is j when below, R is j when not
                        pt = (r_better != below) ? pj : po;
                        // a level only wins with a cost <= err_0 (level 0 pays no penalty): if the right side is better by
                        // less than 2^-45 err_0, fl64(errL + pen) could round onto fl64(errR + pen) and the left point win
                        const uint32_t hg = (uint32_t)__double2hiint(fabs(diff)) + (45u << 20);     // errL - errR = |diff| when R is better
                        slow[k] = slow[k] || (r_better && hg <= hi_d0[k] + (1u << 20));
                    }
                    g[k] = 2 * j + (below ? 1u : 0u);
                    dv[k][lv] = __fmul_rn((float)du, rv[k]);
                    rk[((N - lv) * NE + k) * 256 + tid] = (unsigned short)(((2 * pos + 1) << (N - lv)) - 1);
                    if (WITH_VAL) vl[((N - lv) * NE + k) * 256 + tid] = (float)pt;
                }
            }
        }
        // ---- thresholds in units of b = w / sigma^2, positions in the sorted sweep, guard bands, the "levels lost" column: ONE
        //      straight block for both elements of the lane (their LDS round trips overlap); a sweep point inside a guard
        //      band is only NOTED here (lane masks), the list of such points is made after the block, from dv
        uint64_t fix[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            float Tn[N];
            nb_thresholds(dv[k], Tn);
            float big = dv[k][0];
#pragma unroll
            for (int j = 1; j + 1 < N1; j += 2) big = nb_max3(big, dv[k][j], dv[k][j + 1]);
            fix[k] = __builtin_amdgcn_ballot_w64(!(big < kNbBig) || slow[k]);
            // positions: bucket -> count of sweep points below the bucket -> the one sweep point that may share it
#pragma unroll
            for (int nn = 0; nn < N; ++nn) {
                const int key = min(max(((int)__float_as_uint(Tn[nn]) >> kNbKeyShift) - key0, 0), nkeys - 1);
                const uint32_t c0 = lut[key];
                const float4 nb = rec[c0];
                const float t = Tn[nn];
                const uint32_t a = c0 + (nb.y < t ? 1u : 0u);                     // b_(a-1) < T <= b_(a)
                const float G = __fmul_rn(fmaf(fabsf(t), (float)(nn + 1), dv[k][nn]), 9.5367431640625e-07f);
                const float dist = nb_min3abs(__fsub_rn(t, nb.x), __fsub_rn(t, nb.y), __fsub_rn(t, nb.z));
                fix[k] |= __builtin_amdgcn_ballot_w64(dist <= G);
                atomicAdd(&cnt[((a >> 3) * NE + k) * 256 + tid], 1u << (4u * (a & 7u)));
            }
            fix[k] &= __builtin_amdgcn_ballot_w64(i0 + k < n);
        }
        // ---- rare (about one iteration in ten): which sweep points to re-solve, thresholds again from dv
        uint64_t flags[NE] = {0ull, 0ull};
        if ((fix[0] | fix[1]) != 0ull && !never_flag) {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                if (fix[k] == 0ull) continue;
                const bool mine = (fix[k] >> lane) & 1ull;
                float d2[N1], Tn[N];
#pragma unroll
                for (int j = 0; j < N1; ++j) {                 // opaque copies: nothing below may be hoisted into the hot block
                    float d = dv[k][j];
                    asm volatile("" : "+v"(d));
                    d2[j] = d;
                }
                nb_thresholds(d2, Tn);
                float big = d2[0];
#pragma unroll
                for (int j = 1; j + 1 < N1; j += 2) big = nb_max3(big, d2[j], d2[j + 1]);
                uint64_t fl = (!(big < kNbBig) || slow[k]) ? all_l : 0ull;
#pragma unroll
                for (int nn = 0; nn < N; ++nn) {
                    const int key = min(max(((int)__float_as_uint(Tn[nn]) >> kNbKeyShift) - key0, 0), nkeys - 1);
                    const uint32_t c0 = lut[key];
                    const float4 nb = rec[c0];
                    const uint32_t a = c0 + (nb.y < Tn[nn] ? 1u : 0u);
                    const float G = __fmul_rn(fmaf(fabsf(Tn[nn]), (float)(nn + 1), d2[nn]), 9.5367431640625e-07f);
                    const float dist = nb_min3abs(__fsub_rn(Tn[nn], nb.x), __fsub_rn(Tn[nn], nb.y), __fsub_rn(Tn[nn], nb.z));
                    if (mine && dist <= G) {
                        // the sweep points inside a band are consecutive and next to T: walk outwards from its position
                        // (b_(a-1) < T <= b_(a); rec[l + 1].x = b_(l))
                        for (int l = (int)a - 1; l >= 0 && fabsf(__fsub_rn(rec[l + 1].x, Tn[nn])) <= G; --l) fl |= 1ull << l;
                        for (int l = (int)a; l < L && fabsf(__fsub_rn(rec[l + 1].x, Tn[nn])) <= G; ++l) fl |= 1ull << l;
                    }
                }
                flags[k] = mine ? (fl & all_l) : 0ull;
            }
        }
        // ---- emit the sweep: walk the column, level index r = N - level only grows.  The fields of a word sum to at most 10,
        // so one multiplication by 0x11111111 turns them into their eight running sums (no carries between fields); the
        // running total of the words before enters as an addend of field 0.
        uint32_t cw[CW][NE];
#pragma unroll
        for (int wd = 0; wd < CW; ++wd)
            if (wd * 8 <= L) {
#pragma unroll
                for (int k = 0; k < NE; ++k) cw[wd][k] = atomicExch(&cnt[(wd * NE + k) * 256 + tid], 0u);    // read and clear
            }
        if (full) {
            const char *rkb = reinterpret_cast<const char *>(rk);
            const char *vlb = reinterpret_cast<const char *>(vl);
            uint16_t *oi = out_idx + i0;
            float *ovp = WITH_VAL ? out_val + i0 : nullptr;
            uint32_t base[NE], run[NE] = {0, 0};
#pragma unroll
            for (int k = 0; k < NE; ++k) base[k] = (k * 256 + tid) * 2;            // bits 0 .. 9; the level index goes into bits 10 .. 13
            const int nfull = L >> 3;
            if constexpr (NF > 0 && !WITH_VAL) {
                uint32_t opq = 0;
                asm volatile("" : "+v"(opq));                    // the row numbers are re-read every iteration (not hoisted into 2 NF x 8 SGPRs)
                const uint32_t *pv = reinterpret_cast<const uint32_t *>(perm_s) + opq;
                uint32_t P[NF][NE];
#pragma unroll
                for (int wd = 0; wd < NF; ++wd)
#pragma unroll
                    for (int k = 0; k < NE; ++k) {
                        P[wd][k] = (cw[wd][k] + run[k]) * 0x11111111u;
                        // opaque: the compiler otherwise folds every LEFT shift of P below into a multiplication of its own
                        // (v_mul_lo_u32, quarter rate): 6 NF of them per iteration instead of 2 NF
                        asm volatile("" : "+v"(P[wd][k]));
                        run[k] = P[wd][k] >> 28;
                    }
                const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out_idx, 0, -1 /* 4 GB */, 0x00020000);
                const uint32_t voff2 = (uint32_t)(2 * i0), row_bytes = (uint32_t)(2 * n);
#pragma unroll
                for (int half = 0; half < NF / 2; ++half) {
                    uint32_t v[16], pw4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) pw4[i] = __builtin_amdgcn_readfirstlane(pv[4 * half + i]);
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int wd = 2 * half + w2;
                            uint32_t r2[NE];
#pragma unroll
                            for (int k = 0; k < NE; ++k) {
                                const uint32_t sh = j < 3 ? (P[wd][k] << (10 - 4 * j)) : (P[wd][k] >> (4 * j - 10));
                                r2[k] = *reinterpret_cast<const unsigned short *>(rkb + ((sh & 0x3c00u) | base[k]));
                            }
                            v[8 * w2 + j] = r2[0] | (r2[1] << 16);
                        }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const uint32_t rowi = (pw4[i >> 2] >> (8 * (i & 3))) & 0xffu;
                        if constexpr (BUF) __builtin_amdgcn_raw_buffer_store_b32(v[i], orsrc, voff2, rowi * row_bytes, 2 /* nt */);
                        else __builtin_nontemporal_store(v[i], reinterpret_cast<uint32_t *>(oi + (long)rowi * n));
                    }
                }
            }
#pragma unroll
            for (int wd = (NF > 0 && !WITH_VAL) ? NF : 0; wd < CW - 1; ++wd) {
                if (wd > nfull) break;
                const uint2 pw = reinterpret_cast<const uint2 *>(sw.perm)[wd];      // eight row numbers, one scalar load
                uint32_t P[NE];
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    P[k] = (cw[wd][k] + run[k]) * 0x11111111u;
                    run[k] = P[k] >> 28;
                }
                auto addr = [&](int j, int k) {
                    const uint32_t sh = j < 3 ? (P[k] << (10 - 4 * j)) : (P[k] >> (4 * j - 10));
                    return (sh & 0x3c00u) | base[k];
                };
                auto row = [&](int j) { return ((j < 4 ? pw.x : pw.y) >> (8 * (j & 3))) & 0xffu; };
                if (wd < nfull) {                                                     // a whole word: eight sweep points, no tests
                    uint32_t rank[8][NE];
                    float val[8][NE];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int k = 0; k < NE; ++k) {
                            const uint32_t ad = addr(j, k);
                            rank[j][k] = *reinterpret_cast<const unsigned short *>(rkb + ad);
                            if (WITH_VAL) val[j][k] = *reinterpret_cast<const float *>(vlb + 2 * ad);
                        }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        __builtin_nontemporal_store(rank[j][0] | (rank[j][1] << 16), reinterpret_cast<uint32_t *>(oi + (long)row(j) * n));   // written once, read by a later kernel
                        if (WITH_VAL) *reinterpret_cast<float2 *>(ovp + (long)row(j) * n) = make_float2(val[j][0], val[j][1]);
                    }
                } else {
                    const int lim = L - wd * 8;
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        if (j >= lim) break;
                        const uint32_t a0 = addr(j, 0), a1 = addr(j, 1);
                        *reinterpret_cast<uint32_t *>(oi + (long)row(j) * n) =
                            *reinterpret_cast<const unsigned short *>(rkb + a0) | ((uint32_t)*reinterpret_cast<const unsigned short *>(rkb + a1) << 16);
                        if (WITH_VAL)
                            *reinterpret_cast<float2 *>(ovp + (long)row(j) * n) =
                                make_float2(*reinterpret_cast<const float *>(vlb + 2 * a0), *reinterpret_cast<const float *>(vlb + 2 * a1));
                    }
                }
            }
        } else {
            uint32_t r[NE] = {0, 0};
            for (int l = 0; l < L; ++l) {
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    uint32_t w = 0;
#pragma unroll
                    for (int wd = 0; wd < CW - 1; ++wd) w = (l >> 3) == wd ? cw[wd][k] : w;
                    r[k] += (w >> (4 * (l & 7))) & 15u;
                    if (i0 + k < n) {
                        const long o = (long)sw.perm[l] * n + i0 + k;
                        out_idx[o] = rk[(r[k] * NE + k) * 256 + tid];
                        if (WITH_VAL) out_val[o] = vl[(r[k] * NE + k) * 256 + tid];
                    }
                }
            }
        }
        // ---- sweep points inside a guard band: literal scan, result overwrites the emitted one
        if (__builtin_amdgcn_ballot_w64((flags[0] | flags[1]) != 0ull) != 0ull) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the emitted values have landed before they are replaced
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                while (__builtin_amdgcn_ballot_w64(flags[k] != 0ull) != 0ull) {
                    if (flags[k] != 0ull) {
                        const int l = __builtin_ctzll(flags[k]);
                        flags[k] &= flags[k] - 1ull;
                        const float bl = rec[l].y;
                        const uint32_t rank = exact_rank_scan_nb<N>(tb, (double)m2[k], (double)__fmul_rn(bl, var[k]));
                        const long o = (long)perm_s[l] * n + i0 + k;
                        out_idx[o] = (uint16_t)rank;
                        if (WITH_VAL) {
                            const uint32_t kk = rank + 1;
                            const int tz = __builtin_ctz(kk);
                            out_val[o] = (float)tb[(1 << (N - tz)) - 1 + (kk >> (tz + 1))];
                        }
                    }
                }
            }
        }
    }
}

// Host side of K1nt: sort the sweep by fl32(2 beta), check that the bucket table applies (distinct values at least one
// bucket apart, within 24 octaves).  Returns 1 when the sweep is not eligible (the caller takes the per-beta kernel).
int launch_notebook_hull10(const float *means, const float *stds, int64_t n, const double *codebook, const double *betas,
                           int Lc, uint16_t *oi, float *ov, int vec_ok, int dbg, hipStream_t st) {
    static const bool off = [] { const char *e = getenv("VBQ_NO_HULL"); return e && e[0] == '1'; }();
    if (off || Lc < 1 || Lc > kMaxBetaChunk) return 1;
    NbSweep sw;
    int order[kMaxBetaChunk];
    float b[kMaxBetaChunk];
    for (int i = 0; i < Lc; ++i) { order[i] = i; b[i] = (float)(2.0 * betas[i]); }
    for (int i = 1; i < Lc; ++i)
        for (int j = i; j > 0 && b[order[j]] < b[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    int prev_key = -1;
    for (int i = 0; i < kMaxBetaChunk; ++i) { sw.b[i] = kNbBig; sw.perm[i] = 0; }
    for (int i = 0; i < Lc; ++i) {
        const float v = b[order[i]];
        if (!(v >= 2e-12f && v <= 2e18f)) return 1;
        uint32_t bits;
        memcpy(&bits, &v, 4);
        const int key = (int)(bits >> kNbKeyShift);
        if (key <= prev_key) return 1;
        prev_key = key;
        sw.b[i] = v;
        sw.perm[i] = (unsigned char)order[i];
    }
    uint32_t b0;
    memcpy(&b0, &sw.b[0], 4);
    sw.key0 = (int)(b0 >> kNbKeyShift);
    sw.nkeys = prev_key - sw.key0 + 2;
    sw.L = Lc;
    if (sw.nkeys > kNbKeys) return 1;
    {
        int l = 0;
        for (int k = 0; k < kNbKeys; ++k) {
            while (l < Lc) {
                uint32_t bits;
                memcpy(&bits, &sw.b[l], 4);
                if ((int)(bits >> kNbKeyShift) < sw.key0 + k) ++l; else break;
            }
            sw.lut[k] = (unsigned char)l;
        }
    }
    sw.var_lo = fmaxf(4e-38f / sw.b[0], 1e-30f);
    sw.var_hi = fminf(1e38f / sw.b[Lc - 1], 1e30f);
    int64_t gx = ((n + 1) / 2 + 255) / 256;
    constexpr int rounds = 2;                               // grid = this many times the resident workgroups (measured)
    const int64_t cap = (int64_t)num_cus() * (ov ? 2 : (Lc <= 32 ? 4 : 3)) * rounds;              // persistent grid: every CU's resident workgroups, two rounds
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
#define VBQ_NB_HULL(V, W, F, B)                                                                                       \
    hipLaunchKernelGGL((k_quant_notebook_hull<V, W, F, B>), dim3((unsigned)gx), dim3(256), 0, st, means, stds, (long)n, codebook, sw, oi, \
                       ov, vec_ok, dbg)
    const bool buf = (uint64_t)Lc * 2ull * (uint64_t)n <= 0xffffffffull;       // every index row within 4 GB of the first
    if (ov) {
        if (Lc <= 32) VBQ_NB_HULL(true, 5, 0, false); else VBQ_NB_HULL(true, 9, 0, false);
    } else if (Lc == 32) {                                   // 32 betas: four full words in one straight block
        if (buf) VBQ_NB_HULL(false, 5, 4, true); else VBQ_NB_HULL(false, 5, 4, false);
    } else if ((Lc >> 3) == 6) {                             // 48 .. 55 betas (the notebook's 50): six
        if (buf) VBQ_NB_HULL(false, 9, 6, true); else VBQ_NB_HULL(false, 9, 6, false);
    } else {
        if (Lc <= 32) VBQ_NB_HULL(false, 5, 0, false); else VBQ_NB_HULL(false, 9, 0, false);
    }
#undef VBQ_NB_HULL
    VBQ_CHECK_LAUNCH("quant_notebook_hull");
    return VBQ_OK;
}

template <int N>
int launch_notebook(const float *means, const float *stds, int64_t n, const double *codebook,
                    const double *h_betas, int32_t nb, uint16_t *out_idx, float *out_val, hipStream_t st) {
    for (int l0 = 0; l0 < nb; l0 += kMaxBetaChunk) {
        const int Lc = nb - l0 < kMaxBetaChunk ? nb - l0 : kMaxBetaChunk;
        BetaChunk bc;
        for (int i = 0; i < kMaxBetaChunk; ++i) bc.beta[i] = i < Lc ? h_betas[l0 + i] : 0.0;
        uint16_t *oi = out_idx + (int64_t)l0 * n;
        float *ov = out_val ? out_val + (int64_t)l0 * n : nullptr;
        // pairs: 8-byte loads, 4-byte index stores, 8-byte value stores; unaligned access is enabled on compute queues, so odd
        // n (every other beta plane half a pair off) keeps the paired path
        const int vec_ok = ((reinterpret_cast<uintptr_t>(means) | reinterpret_cast<uintptr_t>(stds)) % 4 == 0) &&
                           (reinterpret_cast<uintptr_t>(oi) % 2 == 0) && (!ov || reinterpret_cast<uintptr_t>(ov) % 4 == 0);
        int64_t gx = ((n + 1) / 2 + 255) / 256;
        if (gx > 2048) gx = 2048;
        if (gx < 1) gx = 1;
        static const bool plain = [] { const char *e = getenv("VBQ_PLAIN_KERNEL"); return e && e[0] == '1'; }();
        static const int dbg = [] { const char *e = getenv("VBQ_FAST_DEBUG"); return e ? atoi(e) : 0; }();
        bool fast_ok = !plain;                      // the tie certificate wants betas in a sane range
        for (int i = 0; i < Lc; ++i) fast_ok = fast_ok && (bc.beta[i] >= 1e-12 && bc.beta[i] <= 1e18);
        static const bool no_pruned = [] { const char *e = getenv("VBQ_NO_PRUNED"); return e && e[0] == '1'; }();
        // The pruned descent stops once best <= w (n + 1): a bound on every deeper level's penalty that holds for w >= 0 only
        // (with a negative beta deeper levels carry SMALLER penalties and can still win): such calls take the literal kernel.
        bool nonneg = true;
        for (int i = 0; i < Lc; ++i) nonneg = nonneg && bc.beta[i] >= 0.0;
        if (!plain && !no_pruned && nonneg && Lc <= 2) {  // one or two betas per call (the notebook's own pattern): pruned descent
            int64_t gp = gx;
            const int64_t capp = (int64_t)num_cus() * 6 * 2;
            if (gp > capp) gp = capp;
            const dim3 grid((unsigned)gp), block(256);
            if (Lc == 1 && ov) hipLaunchKernelGGL((k_quant_notebook_pruned<N, 1, true>), grid, block, 0, st, means, stds, (long)n, codebook, bc, oi, ov, vec_ok);
            else if (Lc == 1) hipLaunchKernelGGL((k_quant_notebook_pruned<N, 1, false>), grid, block, 0, st, means, stds, (long)n, codebook, bc, oi, ov, vec_ok);
            else if (ov) hipLaunchKernelGGL((k_quant_notebook_pruned<N, 2, true>), grid, block, 0, st, means, stds, (long)n, codebook, bc, oi, ov, vec_ok);
            else hipLaunchKernelGGL((k_quant_notebook_pruned<N, 2, false>), grid, block, 0, st, means, stds, (long)n, codebook, bc, oi, ov, vec_ok);
            VBQ_CHECK_LAUNCH("quant_notebook_pruned");
            continue;
        }
        if (fast_ok && N == 10 && Lc >= 6) {           // below that the thresholds cost more than the solves they replace
            const int rc = launch_notebook_hull10(means, stds, n, codebook, bc.beta, Lc, oi, ov, vec_ok, dbg, st);
            if (rc != 1) { if (rc != VBQ_OK) return rc; continue; }
        }
        if (fast_ok)
            hipLaunchKernelGGL((k_quant_notebook_fast<N>), dim3((unsigned)gx), dim3(256), 0, st, means, stds, (long)n,
                               codebook, bc, Lc, oi, ov, vec_ok, dbg);
        else
            hipLaunchKernelGGL((k_quant_notebook<N>), dim3((unsigned)gx), dim3(256), 0, st, means, stds, (long)n, codebook,
                               bc, Lc, oi, ov, vec_ok);
        VBQ_CHECK_LAUNCH("quant_notebook");
    }
    return VBQ_OK;
}

}  // namespace
}  // namespace vbq

extern "C" int vbq_quantize_notebook_f64(const float *d_means, const float *d_stds, int64_t n,
                                         const double *d_codebook_lm, const double *h_betas, int32_t n_beta,
                                         int32_t N, uint16_t *d_out_idx, float *d_out_val, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n >= 0 && n_beta >= 1, VBQ_ERR_INVALID_ARGUMENT, "vbq_quantize_notebook_f64: bad sizes n=%lld n_beta=%d",
                (long long)n, n_beta);
    VBQ_REQUIRE(n == 0 || (d_means && d_stds && d_codebook_lm && h_betas && d_out_idx), VBQ_ERR_INVALID_ARGUMENT,
                "vbq_quantize_notebook_f64: null pointer argument");
    if (n == 0) return VBQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (N) {
        case 10: return launch_notebook<10>(d_means, d_stds, n, d_codebook_lm, h_betas, n_beta, d_out_idx, d_out_val, st);
#ifndef VBQ_ONLY_N10
        case 9: return launch_notebook<9>(d_means, d_stds, n, d_codebook_lm, h_betas, n_beta, d_out_idx, d_out_val, st);
        case 8: return launch_notebook<8>(d_means, d_stds, n, d_codebook_lm, h_betas, n_beta, d_out_idx, d_out_val, st);
        case 7: return launch_notebook<7>(d_means, d_stds, n, d_codebook_lm, h_betas, n_beta, d_out_idx, d_out_val, st);
        case 6: return launch_notebook<6>(d_means, d_stds, n, d_codebook_lm, h_betas, n_beta, d_out_idx, d_out_val, st);
        case 5: return launch_notebook<5>(d_means, d_stds, n, d_codebook_lm, h_betas, n_beta, d_out_idx, d_out_val, st);
        case 4: return launch_notebook<4>(d_means, d_stds, n, d_codebook_lm, h_betas, n_beta, d_out_idx, d_out_val, st);
#endif
        default:
            set_error("vbq_quantize_notebook_f64: max_codepoint_length N=%d not built (have 4 ... 10)", N);
            return VBQ_ERR_UNSUPPORTED;
    }
}

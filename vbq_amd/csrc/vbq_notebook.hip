// K1n: the word-embedding variant of the solve (compress_coordinates,
// word-embeddings/compress-trained-word-embeddings.ipynb:429-443) in its own arithmetic:
//   squared error  fl64(fl64(c - f64(mu))^2)           (ipynb:436, f64 code book minus f32 means)
//   penalty        f64(fl32(fl32(2 beta) * fl32(sigma^2))) * len   (ipynb:438 under NumPy-1.17 casting)
//   argmin over the code book in level-major order, first minimum wins (ipynb:440)
// The notebook scores all 2047 points; only the two neighbours of mu on each bit level can
// win (rounding is monotone, so a farther point of the same level never scores lower), so
// the same 21-candidate descent as K1 is used, scanned in level-major order.
#include <stdlib.h>

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

template <int N>
int launch_notebook(const float *means, const float *stds, int64_t n, const double *codebook,
                    const double *h_betas, int32_t nb, uint16_t *out_idx, float *out_val, hipStream_t st) {
    for (int l0 = 0; l0 < nb; l0 += kMaxBetaChunk) {
        const int Lc = nb - l0 < kMaxBetaChunk ? nb - l0 : kMaxBetaChunk;
        BetaChunk bc;
        for (int i = 0; i < kMaxBetaChunk; ++i) bc.beta[i] = i < Lc ? h_betas[l0 + i] : 0.0;
        uint16_t *oi = out_idx + (int64_t)l0 * n;
        float *ov = out_val ? out_val + (int64_t)l0 * n : nullptr;
        const int vec_ok = ((reinterpret_cast<uintptr_t>(means) | reinterpret_cast<uintptr_t>(stds)) % 8 == 0) &&
                           (reinterpret_cast<uintptr_t>(oi) % 4 == 0) && (n % 2 == 0 || nb == 1) &&
                           (!ov || reinterpret_cast<uintptr_t>(ov) % 8 == 0);
        int64_t gx = ((n + 1) / 2 + 255) / 256;
        if (gx > 2048) gx = 2048;
        if (gx < 1) gx = 1;
        static const bool plain = [] { const char *e = getenv("VBQ_PLAIN_KERNEL"); return e && e[0] == '1'; }();
        static const int dbg = [] { const char *e = getenv("VBQ_FAST_DEBUG"); return e ? atoi(e) : 0; }();
        bool fast_ok = !plain;                      // the tie certificate wants betas in a sane range
        for (int i = 0; i < Lc; ++i) fast_ok = fast_ok && (bc.beta[i] >= 1e-12 && bc.beta[i] <= 1e18);
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

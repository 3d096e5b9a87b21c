// K1c: utils.batch_quantize_indep_dims (utils.py:363-423) for caller-built candidate
// tensors P, L of shape M x (B*K).  This is the reference's materialised formulation --
// M reads per element -- kept as a drop-in for callers that bring their own candidates;
// the fused K1 (vbq_quantize.hip) is the hot path.
#include "vbq_common.h"

namespace vbq {
namespace {

constexpr int kLamChunk = 8;
struct Lam8 {
    double lam[kLamChunk];
};

template <bool F64>
__global__ void __launch_bounds__(256)
k_argmax_candidates(const float *__restrict__ P, const float *__restrict__ len, long len_lambda_stride,
                    const float *__restrict__ mu, const float *__restrict__ sg, long E, Lam8 lc, int Lc, int M,
                    int32_t *__restrict__ out_j, float *__restrict__ out_z, float *__restrict__ out_b) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        const float m = mu[e], s = sg[e];
        float bestf[kLamChunk];
        double bestd[kLamChunk];
        int bj[kLamChunk];
#pragma unroll
        for (int l = 0; l < kLamChunk; ++l) { bj[l] = 0; bestf[l] = 0.f; bestd[l] = 0.0; }
        for (int j = 0; j < M; ++j) {
            const float d = neg_half_sq_err(P[(long)j * E + e], m, s);
#pragma unroll
            for (int l = 0; l < kLamChunk; ++l) {
                if (l < Lc) {
                    const float ln = len[l * len_lambda_stride + (long)j * E + e];
                    bool up;
                    if (F64) {
                        const double sc = __dsub_rn((double)d, __dmul_rn(lc.lam[l], (double)ln));
                        up = j == 0 || sc > bestd[l];
                        bestd[l] = up ? sc : bestd[l];
                    } else {
                        const float sc = __fsub_rn(d, __fmul_rn((float)lc.lam[l], ln));
                        up = j == 0 || sc > bestf[l];
                        bestf[l] = up ? sc : bestf[l];
                    }
                    bj[l] = up ? j : bj[l];
                }
            }
        }
#pragma unroll
        for (int l = 0; l < kLamChunk; ++l) {
            if (l < Lc) {
                const long o = (long)l * E + e;
                if (out_j) out_j[o] = bj[l];
                if (out_z) out_z[o] = P[(long)bj[l] * E + e];
                if (out_b) out_b[o] = len[l * len_lambda_stride + (long)bj[l] * E + e];
            }
        }
    }
}

}  // namespace
}  // namespace vbq

extern "C" int vbq_argmax_candidates_f32(const float *d_P, const float *d_len, int32_t len_per_lambda,
                                         const float *d_mu, const float *d_sigma, int64_t n_elems,
                                         const double *h_lambdas, int32_t n_lambda, int32_t M, int32_t mode,
                                         int32_t *d_out_j, float *d_out_zhat, float *d_out_bits, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(d_P && d_len && d_mu && d_sigma && h_lambdas, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_argmax_candidates_f32: null pointer argument");
    VBQ_REQUIRE(n_elems >= 0 && n_lambda >= 1 && M >= 1 && M <= 65535, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_argmax_candidates_f32: bad sizes n_elems=%lld n_lambda=%d M=%d (1 <= M <= 65535)",
                (long long)n_elems, n_lambda, M);
    VBQ_REQUIRE(mode == VBQ_MODE_F32 || mode == VBQ_MODE_F64_SCORE, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_argmax_candidates_f32: unknown mode %d", mode);
    if (n_elems == 0) return VBQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int64_t gx = (n_elems + 255) / 256;
    if (gx > 4096) gx = 4096;
    const long lstride = len_per_lambda ? (long)M * n_elems : 0;
    for (int l0 = 0; l0 < n_lambda; l0 += kLamChunk) {
        const int Lc = n_lambda - l0 < kLamChunk ? n_lambda - l0 : kLamChunk;
        Lam8 lc;
        for (int i = 0; i < kLamChunk; ++i) lc.lam[i] = i < Lc ? h_lambdas[l0 + i] : 0.0;
        const float *len = d_len + (int64_t)l0 * lstride;
        int32_t *oj = d_out_j ? d_out_j + (int64_t)l0 * n_elems : nullptr;
        float *oz = d_out_zhat ? d_out_zhat + (int64_t)l0 * n_elems : nullptr;
        float *ob = d_out_bits ? d_out_bits + (int64_t)l0 * n_elems : nullptr;
        if (mode == VBQ_MODE_F32)
            hipLaunchKernelGGL((k_argmax_candidates<false>), dim3((unsigned)gx), dim3(256), 0, st, d_P, len, lstride,
                               d_mu, d_sigma, (long)n_elems, lc, Lc, (int)M, oj, oz, ob);
        else
            hipLaunchKernelGGL((k_argmax_candidates<true>), dim3((unsigned)gx), dim3(256), 0, st, d_P, len, lstride,
                               d_mu, d_sigma, (long)n_elems, lc, Lc, (int)M, oj, oz, ob);
        VBQ_CHECK_LAUNCH("argmax_candidates");
    }
    return VBQ_OK;
}

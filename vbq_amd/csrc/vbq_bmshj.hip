// K4: BMSHJ2018 prior (img-compression/learned_prior.py) -- elementwise CDF / analytic PDF /
// log-PDF and one masked bisection update of inverse_cdf.  Build-time kernels (2047*C points
// per table), so they are written for clarity; the per-channel 43 effective parameters are
// read through the vector L1.
#include "vbq_common.h"

namespace vbq {
namespace {

struct CdfPdf {
    float cdf, pdf;
};

// learned_prior.py:88-107 (logits), :140 (sigmoid), :277-321 (Jacobian chain); dims (1,3,3,3,1).
__device__ __forceinline__ CdfPdf bmshj_eval(const float *__restrict__ P, float x) {
    float h[3], v[3];
    // layer 0: 3x1
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float M = P[r], b = P[3 + r], f = P[6 + r];
        float a = __fadd_rn(__fmul_rn(M, x), b);
        const float t = tanhf(a);
        a = __fadd_rn(a, __fmul_rn(f, t));
        const float g = __fadd_rn(1.0f, __fmul_rn(f, __fsub_rn(1.0f, __fmul_rn(t, t))));
        h[r] = a;
        v[r] = __fmul_rn(g, M);
    }
    // layers 1, 2: 3x3
#pragma unroll
    for (int layer = 0; layer < 2; ++layer) {
        const float *Q = P + 9 + 15 * layer;
        float hn[3], vn[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float m0 = Q[3 * r], m1 = Q[3 * r + 1], m2 = Q[3 * r + 2];
            const float b = Q[9 + r], f = Q[12 + r];
            float a = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m0, h[0]), __fmul_rn(m1, h[1])), __fmul_rn(m2, h[2])), b);
            const float t = tanhf(a);
            a = __fadd_rn(a, __fmul_rn(f, t));
            const float g = __fadd_rn(1.0f, __fmul_rn(f, __fsub_rn(1.0f, __fmul_rn(t, t))));
            const float jv = __fadd_rn(__fadd_rn(__fmul_rn(m0, v[0]), __fmul_rn(m1, v[1])), __fmul_rn(m2, v[2]));
            hn[r] = a;
            vn[r] = __fmul_rn(g, jv);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) { h[r] = hn[r]; v[r] = vn[r]; }
    }
    // layer 3: 1x3 + sigmoid
    const float *Q = P + 39;
    const float lg = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(Q[0], h[0]), __fmul_rn(Q[1], h[1])), __fmul_rn(Q[2], h[2])), Q[3]);
    const float jv = __fadd_rn(__fadd_rn(__fmul_rn(Q[0], v[0]), __fmul_rn(Q[1], v[1])), __fmul_rn(Q[2], v[2]));
    CdfPdf o;
    o.cdf = 1.0f / (1.0f + expf(-lg));
    o.pdf = __fmul_rn(__fmul_rn(o.cdf, __fsub_rn(1.0f, o.cdf)), jv);
    return o;
}

__global__ void __launch_bounds__(256)
k_bmshj_cdf_pdf(const float *__restrict__ params, const float *__restrict__ x, long E, int C,
                float *__restrict__ cdf, float *__restrict__ pdf, float *__restrict__ logpdf) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const CdfPdf o = bmshj_eval(params + (long)c * VBQ_BMSHJ_PARAMS_PER_CHANNEL, x[e]);
        if (cdf) cdf[e] = o.cdf;
        if (pdf) pdf[e] = o.pdf;
        if (logpdf) logpdf[e] = logf(__fadd_rn(o.pdf, 1e-10f));     // learned_prior.py:242
    }
}

// learned_prior.py:199-209 for one iteration; the caller applies the stopping rule of :210-211
// from flags[0] (number of non-zero mid values) and flags[1] (min bracket width, as u32 bits).
__global__ void __launch_bounds__(256)
k_bmshj_icdf_step(const float *__restrict__ params, const float *__restrict__ xi, long E, int C,
                  float *__restrict__ left, float *__restrict__ right, float *__restrict__ mid,
                  unsigned int *__restrict__ flags) {
    unsigned int nz = 0, wmin = 0x7f800000u;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        float lo = left[e], hi = right[e];
        const float m = __fmul_rn(0.5f, __fadd_rn(lo, hi));
        const float val = __fsub_rn(bmshj_eval(params + (long)c * VBQ_BMSHJ_PARAMS_PER_CHANNEL, m).cdf, xi[e]);
        if (val < 0.0f) lo = m;
        if (val > 0.0f) hi = m;
        left[e] = lo;
        right[e] = hi;
        mid[e] = m;
        nz += (val != 0.0f) ? 1u : 0u;
        const float w = __fsub_rn(hi, lo);
        const unsigned int wb = w > 0.0f ? __float_as_uint(w) : 0u;
        wmin = wb < wmin ? wb : wmin;
    }
    if (nz) atomicAdd(&flags[0], nz);
    atomicMin(&flags[1], wmin);
}

}  // namespace
}  // namespace vbq

extern "C" int vbq_bmshj_cdf_pdf_f32(const float *d_params, const float *d_x, int64_t n_rows, int32_t n_ch,
                                     float *d_cdf, float *d_pdf, float *d_logpdf, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(d_params && d_x, VBQ_ERR_INVALID_ARGUMENT, "vbq_bmshj_cdf_pdf_f32: null pointer argument");
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1, VBQ_ERR_INVALID_ARGUMENT, "vbq_bmshj_cdf_pdf_f32: bad sizes");
    const int64_t E = n_rows * (int64_t)n_ch;
    if (E == 0) return VBQ_OK;
    int64_t gx = (E + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_bmshj_cdf_pdf, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_params,
                       d_x, (long)E, (int)n_ch, d_cdf, d_pdf, d_logpdf);
    VBQ_CHECK_LAUNCH("bmshj_cdf_pdf");
    return VBQ_OK;
}

extern "C" int vbq_bmshj_icdf_step_f32(const float *d_params, const float *d_xi, int64_t n_rows, int32_t n_ch,
                                       float *d_left, float *d_right, float *d_mid, uint32_t *d_flags,
                                       void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(d_params && d_xi && d_left && d_right && d_mid && d_flags, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_bmshj_icdf_step_f32: null pointer argument");
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1, VBQ_ERR_INVALID_ARGUMENT, "vbq_bmshj_icdf_step_f32: bad sizes");
    const int64_t E = n_rows * (int64_t)n_ch;
    if (E == 0) return VBQ_OK;
    int64_t gx = (E + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_bmshj_icdf_step, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       d_params, d_xi, (long)E, (int)n_ch, d_left, d_right, d_mid, d_flags);
    VBQ_CHECK_LAUNCH("bmshj_icdf_step");
    return VBQ_OK;
}

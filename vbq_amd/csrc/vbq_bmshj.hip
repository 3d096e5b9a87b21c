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
                float *__restrict__ cdf, float *__restrict__ pdf, float *__restrict__ logpdf, int staged) {
    extern __shared__ float sp[];                             // the parameter table, staged (see k_bmshj_icdf_chain)
    if (staged) {
        for (int i = threadIdx.x; i < C * VBQ_BMSHJ_PARAMS_PER_CHANNEL; i += blockDim.x) sp[i] = params[i];
        __syncthreads();
    }
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const CdfPdf o = staged ? bmshj_eval(sp + c * VBQ_BMSHJ_PARAMS_PER_CHANNEL, x[e])
                                : bmshj_eval(params + (long)c * VBQ_BMSHJ_PARAMS_PER_CHANNEL, x[e]);
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

// The same update as step j of a CHAIN of steps enqueued together: it runs only while the stopping rule of learned_prior.py:210-211
// has not been met by the step before it -- flags[j] = what step j - 1 accumulated: { #(f(mid) != 0), bits of the minimum bracket
// width } --, and accumulates its own pair into flags[j + 1].  A skipped step leaves its pair at the initial (0, +inf), which reads
// "done" to the next one: the rule propagates down the chain without the host, and mid keeps the value of the last step that ran
// (what the reference's `break` returns).  flags[0] = (1, +inf): the first step always runs.
__global__ void __launch_bounds__(256)
k_bmshj_icdf_chain(const float *__restrict__ params, const float *__restrict__ xi, long E, int C,
                   float *__restrict__ left, float *__restrict__ right, float *__restrict__ mid,
                   const unsigned int *__restrict__ prev, unsigned int *__restrict__ cur, float tol, int staged) {
    if (prev[0] == 0u || __uint_as_float(prev[1]) <= tol) return;
    // every lane of a wave sits on another channel: read straight from memory, its 43 parameters are 43 gathers of 4 bytes per
    // point (0.22 ms per step at 2047 x 256 points); staged in LDS -- rows of 43 words, an odd stride: no bank conflicts -- the
    // step is arithmetic
    extern __shared__ float sp[];
    if (staged) {
        for (int i = threadIdx.x; i < C * VBQ_BMSHJ_PARAMS_PER_CHANNEL; i += blockDim.x) sp[i] = params[i];
        __syncthreads();
    }
    unsigned int nz = 0, wmin = 0x7f800000u;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        float lo = left[e], hi = right[e];
        const float m = __fmul_rn(0.5f, __fadd_rn(lo, hi));
        const float cdf = staged ? bmshj_eval(sp + c * VBQ_BMSHJ_PARAMS_PER_CHANNEL, m).cdf
                                 : bmshj_eval(params + (long)c * VBQ_BMSHJ_PARAMS_PER_CHANNEL, m).cdf;
        const float val = __fsub_rn(cdf, xi[e]);
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
    if (nz) atomicAdd(&cur[0], nz);
    atomicMin(&cur[1], wmin);
}

__global__ void k_bmshj_icdf_chain_init(unsigned int *__restrict__ flags, int n_steps, int first) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j <= n_steps; j += gridDim.x * blockDim.x) {
        if (j == 0 && !first) continue;                         // a continued chain keeps what its last step accumulated
        flags[2 * j] = j == 0 ? 1u : 0u;
        flags[2 * j + 1] = 0x7f800000u;
    }
}

}  // namespace
}  // namespace vbq

extern "C" int vbq_bmshj_icdf_chain_f32(const float *d_params, const float *d_xi, int64_t n_rows, int32_t n_ch,
                                        float *d_left, float *d_right, float *d_mid, uint32_t *d_flags, int32_t n_steps,
                                        float tol, int32_t first, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(d_params && d_xi && d_left && d_right && d_mid && d_flags, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_bmshj_icdf_chain_f32: null pointer argument");
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_steps >= 1 && n_steps <= 4096, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_bmshj_icdf_chain_f32: bad sizes (n_steps must be 1..4096)");
    const int64_t E = n_rows * (int64_t)n_ch;
    if (E == 0) return VBQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int64_t gx = (E + 255) / 256;
    const size_t lds = (size_t)n_ch * VBQ_BMSHJ_PARAMS_PER_CHANNEL * sizeof(float);
    const int staged = lds <= 48 * 1024 && gx >= 8;           // (few points: the staging would cost more than the gathers)
    if (staged) {
        const int64_t cap = 2 * (int64_t)num_cus();             // every workgroup stages the table once: few, long-lived ones
        if (gx > cap) gx = cap;
    } else if (gx > 4096) {
        gx = 4096;
    }
    hipLaunchKernelGGL(k_bmshj_icdf_chain_init, dim3((unsigned)((n_steps + 256) / 256)), dim3(256), 0, st, d_flags, (int)n_steps,
                       (int)(first != 0));
    for (int j = 0; j < n_steps; ++j)
        hipLaunchKernelGGL(k_bmshj_icdf_chain, dim3((unsigned)gx), dim3(256), staged ? lds : 0, st, d_params, d_xi, (long)E,
                           (int)n_ch, d_left, d_right, d_mid, d_flags + 2 * j, d_flags + 2 * (j + 1), tol, staged);
    VBQ_CHECK_LAUNCH("bmshj_icdf_chain");
    return VBQ_OK;
}

extern "C" int vbq_bmshj_cdf_pdf_f32(const float *d_params, const float *d_x, int64_t n_rows, int32_t n_ch,
                                     float *d_cdf, float *d_pdf, float *d_logpdf, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(d_params && d_x, VBQ_ERR_INVALID_ARGUMENT, "vbq_bmshj_cdf_pdf_f32: null pointer argument");
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1, VBQ_ERR_INVALID_ARGUMENT, "vbq_bmshj_cdf_pdf_f32: bad sizes");
    const int64_t E = n_rows * (int64_t)n_ch;
    if (E == 0) return VBQ_OK;
    int64_t gx = (E + 255) / 256;
    const size_t lds = (size_t)n_ch * VBQ_BMSHJ_PARAMS_PER_CHANNEL * sizeof(float);
    const int staged = lds <= 48 * 1024 && gx >= 8;
    if (staged) {
        const int64_t cap = 2 * (int64_t)num_cus();
        if (gx > cap) gx = cap;
    } else if (gx > 4096) {
        gx = 4096;
    }
    hipLaunchKernelGGL(k_bmshj_cdf_pdf, dim3((unsigned)gx), dim3(256), staged ? lds : 0, reinterpret_cast<hipStream_t>(stream), d_params,
                       d_x, (long)E, (int)n_ch, d_cdf, d_pdf, d_logpdf, staged);
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

// ------------------------------------------------------------------------------------------
// Fit step of the prior (learned_prior.py:363-465, the "next" row f1 of SURVEY 8f):
//   loss = -mean(log(pdf(x) + 1e-10)),  full batch.
// One pass accumulates, per channel, sum(-log(pdf+1e-10)) and its gradient with respect to the
// 43 EFFECTIVE parameters by hand-written reverse mode through the value path (h) and the
// tangent path (v = dh/dx) of the 4-layer map; the host applies the softplus / tanh chain rule
// and Adam (43*C numbers).  x is read as channel-major planes so that a workgroup has its 43
// parameters in scalar registers and every thread keeps 44 private partial sums.
// ------------------------------------------------------------------------------------------
namespace vbq {
namespace {

constexpr int kNP = VBQ_BMSHJ_PARAMS_PER_CHANNEL;

__device__ __forceinline__ void bmshj_nll_grad(const float *__restrict__ P, float x, float (&g)[kNP + 1]) {
    // ---- forward, keeping what the backward pass needs
    float a[3][3], t[3][3], gg[3][3], u[3][3], h[3][3], v[3][3];     // [layer][unit]
    // layer 0 (3x1): h_-1 = x, v_-1 = 1
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float M = P[r], b = P[3 + r], f = P[6 + r];
        a[0][r] = M * x + b;
        u[0][r] = M;
        t[0][r] = tanhf(a[0][r]);
        gg[0][r] = 1.0f + f * (1.0f - t[0][r] * t[0][r]);
        h[0][r] = a[0][r] + f * t[0][r];
        v[0][r] = gg[0][r] * u[0][r];
    }
#pragma unroll
    for (int L = 1; L <= 2; ++L) {
        const float *Q = P + 9 + 15 * (L - 1);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float f = Q[12 + r];
            a[L][r] = Q[3 * r] * h[L - 1][0] + Q[3 * r + 1] * h[L - 1][1] + Q[3 * r + 2] * h[L - 1][2] + Q[9 + r];
            u[L][r] = Q[3 * r] * v[L - 1][0] + Q[3 * r + 1] * v[L - 1][1] + Q[3 * r + 2] * v[L - 1][2];
            t[L][r] = tanhf(a[L][r]);
            gg[L][r] = 1.0f + f * (1.0f - t[L][r] * t[L][r]);
            h[L][r] = a[L][r] + f * t[L][r];
            v[L][r] = gg[L][r] * u[L][r];
        }
    }
    const float *Q3 = P + 39;
    const float a3 = Q3[0] * h[2][0] + Q3[1] * h[2][1] + Q3[2] * h[2][2] + Q3[3];
    const float u3 = Q3[0] * v[2][0] + Q3[1] * v[2][1] + Q3[2] * v[2][2];
    const float s = 1.0f / (1.0f + expf(-a3));
    const float sp = s * (1.0f - s);
    const float pdf = sp * u3;
    const float pe = pdf + 1e-10f;
    g[kNP] += -logf(pe);
    // ---- backward of L = -log(pdf + eps)
    const float pb = -1.0f / pe;
    float ab = pb * u3 * sp * (1.0f - 2.0f * s);       // dL/da3
    float ub = pb * sp;                                 // dL/du3
    float hb[3], vb[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        g[39 + d] += ab * h[2][d] + ub * v[2][d];
        hb[d] = Q3[d] * ab;
        vb[d] = Q3[d] * ub;
    }
    g[42] += ab;
#pragma unroll
    for (int L = 2; L >= 1; --L) {
        const float *Q = P + 9 + 15 * (L - 1);
        float *G = g + 9 + 15 * (L - 1);
        float abv[3], ubv[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float f = Q[12 + r];
            const float omt = 1.0f - t[L][r] * t[L][r];
            G[12 + r] += hb[r] * t[L][r] + vb[r] * u[L][r] * omt;
            abv[r] = hb[r] * gg[L][r] + vb[r] * u[L][r] * f * (-2.0f * t[L][r] * omt);
            ubv[r] = vb[r] * gg[L][r];
            G[9 + r] += abv[r];
#pragma unroll
            for (int d = 0; d < 3; ++d) G[3 * r + d] += abv[r] * h[L - 1][d] + ubv[r] * v[L - 1][d];
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            hb[d] = Q[d] * abv[0] + Q[3 + d] * abv[1] + Q[6 + d] * abv[2];
            vb[d] = Q[d] * ubv[0] + Q[3 + d] * ubv[1] + Q[6 + d] * ubv[2];
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float f = P[6 + r];
        const float omt = 1.0f - t[0][r] * t[0][r];
        g[6 + r] += hb[r] * t[0][r] + vb[r] * u[0][r] * omt;
        const float a0b = hb[r] * gg[0][r] + vb[r] * u[0][r] * f * (-2.0f * t[0][r] * omt);
        const float u0b = vb[r] * gg[0][r];
        g[3 + r] += a0b;
        g[r] += a0b * x + u0b;             // h_-1 = x, v_-1 = 1
    }
}

__global__ void __launch_bounds__(256)
k_bmshj_nll_grad(const float *__restrict__ params, const float *__restrict__ x_cb, long n_rows,
                 double *__restrict__ out) {
    const int c = blockIdx.y;
    __shared__ float P[kNP + 1];
    __shared__ double red[4][kNP + 1];
    if (threadIdx.x < kNP) P[threadIdx.x] = params[(long)c * kNP + threadIdx.x];
    __syncthreads();
    float Pl[kNP];
#pragma unroll
    for (int i = 0; i < kNP; ++i) Pl[i] = P[i];
    float g[kNP + 1];
#pragma unroll
    for (int i = 0; i <= kNP; ++i) g[i] = 0.0f;
    const float *src = x_cb + (long)c * n_rows;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (long)gridDim.x * blockDim.x)
        bmshj_nll_grad(Pl, src[r], g);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i <= kNP; ++i) {
        double s = (double)g[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) red[w][i] = s;
    }
    __syncthreads();
    if (threadIdx.x <= kNP) {
        const int i = threadIdx.x;
        atomicAdd(&out[(long)c * (kNP + 1) + i], (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]));
    }
}

}  // namespace
}  // namespace vbq

extern "C" int vbq_bmshj_nll_grad_f32(const float *d_params, const float *d_x_cb, int64_t n_rows, int32_t n_ch,
                                      double *d_out, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_ch <= 65535, VBQ_ERR_INVALID_ARGUMENT, "vbq_bmshj_nll_grad_f32: bad sizes");
    if (n_rows == 0) return VBQ_OK;
    VBQ_REQUIRE(d_params && d_x_cb && d_out, VBQ_ERR_INVALID_ARGUMENT, "vbq_bmshj_nll_grad_f32: null pointer argument");
    int64_t gx = (n_rows + 256 * 8 - 1) / (256 * 8);
    const int64_t cap = 4096 / n_ch + 1;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(k_bmshj_nll_grad, dim3((unsigned)gx, (unsigned)n_ch), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), d_params, d_x_cb, (long)n_rows, d_out);
    VBQ_CHECK_LAUNCH("bmshj_nll_grad");
    return VBQ_OK;
}

// f3: the comparison quantizers of img-compression/quantizer.py:259-333 as
// elementwise kernels -- uniform grid (UniformQuantizer.quantize / the index pass of .fit) and
// nearest code point (KmeansQuantizer.quantize, scipy.cluster.vq.vq).  Same arithmetic as the
// NumPy / SciPy code: f32 sub, IEEE div, floor, clip for the grid; f64 squared distance and
// first minimum in code-book order for vq.  Optional fused bincount (64-bit atomics).
#include "vbq_common.h"

namespace vbq {
namespace {

__global__ void __launch_bounds__(256)
k_uniform_quantize(const float *__restrict__ x, long n, float mn, float delta, float offset, int levels,
                   float *__restrict__ out_i, float *__restrict__ out_q, unsigned long long *__restrict__ counts) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        float I = floorf(__fdiv_rn(__fsub_rn(x[e], mn), delta));                 // quantizer.py:280,295
        I = fminf(fmaxf(I, 0.0f), (float)(levels - 1));
        if (out_i) out_i[e] = I;
        if (out_q) out_q[e] = __fadd_rn(offset, __fmul_rn(delta, I));             // :297
        if (counts) atomicAdd(&counts[(int)I], 1ULL);
    }
}

__global__ void __launch_bounds__(256)
k_nearest_code(const float *__restrict__ x, long n, const double *__restrict__ codes, int K,
               int *__restrict__ out_i, double *__restrict__ out_q, unsigned long long *__restrict__ counts) {
    extern __shared__ double cb[];
    for (int i = threadIdx.x; i < K; i += blockDim.x) cb[i] = codes[i];
    __syncthreads();
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const double v = (double)x[e];
        int best = 0;
        double bd = 0.0;
        for (int k = 0; k < K; ++k) {
            const double d = __dsub_rn(v, cb[k]);
            const double d2 = __dmul_rn(d, d);
            if (k == 0 || d2 < bd) { bd = d2; best = k; }
        }
        if (out_i) out_i[e] = best;
        if (out_q) out_q[e] = cb[best];
        if (counts) atomicAdd(&counts[best], 1ULL);
    }
}

}  // namespace
}  // namespace vbq

extern "C" int vbq_uniform_quantize_f32(const float *d_x, int64_t n, float min, float delta, float offset,
                                        int32_t levels, float *d_out_index, float *d_out_value, int64_t *d_counts,
                                        void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n >= 0 && levels >= 1 && delta > 0.0f, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_uniform_quantize_f32: bad arguments n=%lld levels=%d delta=%g", (long long)n, levels, (double)delta);
    if (n == 0) return VBQ_OK;
    VBQ_REQUIRE(d_x, VBQ_ERR_INVALID_ARGUMENT, "vbq_uniform_quantize_f32: null input");
    int64_t gx = (n + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_uniform_quantize, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_x,
                       (long)n, min, delta, offset, (int)levels, d_out_index, d_out_value,
                       reinterpret_cast<unsigned long long *>(d_counts));
    VBQ_CHECK_LAUNCH("uniform_quantize");
    return VBQ_OK;
}

extern "C" int vbq_nearest_code_f64(const float *d_x, int64_t n, const double *d_codes, int32_t n_codes,
                                    int32_t *d_out_index, double *d_out_value, int64_t *d_counts, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n >= 0 && n_codes >= 1 && n_codes <= 8192, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_nearest_code_f64: bad arguments n=%lld n_codes=%d (<= 8192)", (long long)n, n_codes);
    if (n == 0) return VBQ_OK;
    VBQ_REQUIRE(d_x && d_codes, VBQ_ERR_INVALID_ARGUMENT, "vbq_nearest_code_f64: null input");
    int64_t gx = (n + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_nearest_code, dim3((unsigned)gx), dim3(256), sizeof(double) * n_codes,
                       reinterpret_cast<hipStream_t>(stream), d_x, (long)n, d_codes, (int)n_codes, d_out_index, d_out_value,
                       reinterpret_cast<unsigned long long *>(d_counts));
    VBQ_CHECK_LAUNCH("nearest_code");
    return VBQ_OK;
}

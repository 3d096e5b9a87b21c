// K2 histogram, K3 moments, rank-index table lookup.   gfx950 / CDNA4 only.
//
// K2 replaces the per-channel np.bincount of quantizer.py:104-105,138-140 and the Counter
// of ipynb:453.  One workgroup privatises the bins of one (lambda, channel[-group]) in LDS
// (u32), streams its share of the u16 indices with 16-B loads, and flushes the non-zero
// bins with one 64-bit global atomic each.  Integer adds: the result does not depend on
// the order of arrival, the grid shape or the number of GPUs.
#include "vbq_common.h"

namespace vbq {
namespace {

constexpr int kHistThreads = 256;

// All indices of the workgroup belong to one channel: [l][c][n_per_ch] contiguous.
template <int N>
__global__ void __launch_bounds__(kHistThreads)
k_hist_flat(const uint16_t *__restrict__ idx, long n_per_ch, int C, long E,
            unsigned long long *__restrict__ counts, int vec_ok) {
    constexpr int T = table_size(N);
    __shared__ unsigned int h[T + 1];
    const int c = blockIdx.y, l = blockIdx.z;
    for (int i = threadIdx.x; i <= T; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint16_t *src = idx + (long)l * E + (long)c * n_per_ch;
    const long noct = vec_ok ? (n_per_ch >> 3) : 0;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < noct; q += (long)gridDim.x * blockDim.x) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + q * 8);
        const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicAdd(&h[w[k] & 0xffffu], 1u);
            atomicAdd(&h[w[k] >> 16], 1u);
        }
    }
    for (long i = noct * 8 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_per_ch;
         i += (long)gridDim.x * blockDim.x)
        atomicAdd(&h[src[i]], 1u);
    __syncthreads();
    unsigned long long *dst = counts + ((long)l * C + c) * T;
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        const unsigned int v = h[i];
        if (v) atomicAdd(&dst[i], (unsigned long long)v);
    }
}

// Channel-last [rows][C], C > 1: a workgroup owns 16 consecutive channels of one lambda.
constexpr int kHistTiledThreads = 1024;

template <int N>
__global__ void __launch_bounds__(kHistTiledThreads)
k_hist_tiled(const uint16_t *__restrict__ idx, long n_rows, int C, long E,
             unsigned long long *__restrict__ counts) {
    constexpr int T = table_size(N);
    constexpr int TS = T + 2;
    extern __shared__ unsigned int hs[];
    const int c0 = blockIdx.y * kTileChannels, l = blockIdx.z;
    const int ncg = min(kTileChannels, C - c0);
    for (int i = threadIdx.x; i < kTileChannels * TS; i += blockDim.x) hs[i] = 0;
    __syncthreads();
    const int cl = threadIdx.x & (kTileChannels - 1);
    const int slot = threadIdx.x >> 4;
    const uint16_t *src = idx + (long)l * E;
    if (cl < ncg) {
        for (long r = (long)blockIdx.x * 64 + slot; r < n_rows; r += (long)gridDim.x * 64)
            atomicAdd(&hs[cl * TS + src[r * C + c0 + cl]], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncg * T; i += blockDim.x) {
        const int ch = i / T, s = i - ch * T;
        const unsigned int v = hs[ch * TS + s];
        if (v) atomicAdd(&counts[((long)l * C + c0 + ch) * T + s], (unsigned long long)v);
    }
}

template <int N>
int launch_hist(const uint16_t *idx, int64_t n_rows, int32_t n_ch, int32_t layout, int32_t L,
                unsigned long long *counts, hipStream_t st) {
    const int64_t E = n_rows * (int64_t)n_ch;
    const bool flat = (n_ch == 1) || (layout == VBQ_LAYOUT_CB);
    if (flat) {
        const int64_t n_per_ch = (n_ch == 1) ? E : n_rows;
        const int vec_ok = (reinterpret_cast<uintptr_t>(idx) % 16 == 0) && (n_per_ch % 8 == 0 || (n_ch == 1 && L == 1)) &&
                           (E % 8 == 0 || L == 1);
        int64_t gx = (n_per_ch / 8 + kHistThreads - 1) / kHistThreads;
        int64_t cap = (int64_t)2048 / ((int64_t)n_ch * L) + 1;
        if (gx > cap) gx = cap;
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL((k_hist_flat<N>), dim3((unsigned)gx, (unsigned)n_ch, (unsigned)L), dim3(kHistThreads), 0, st,
                           idx, (long)n_per_ch, (int)n_ch, (long)E, counts, vec_ok);
        VBQ_CHECK_LAUNCH("hist_flat");
    } else {
        constexpr int T = table_size(N);
        const size_t lds = sizeof(unsigned int) * kTileChannels * (T + 2);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hist_tiled<N>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
                return VBQ_ERR_LAUNCH;
            }
            attr_set = true;
        }
        const int groups = (n_ch + kTileChannels - 1) / kTileChannels;
        int64_t gx = 512 / ((int64_t)groups * L) + 1;
        const int64_t iters = (n_rows + 63) / 64;
        if (gx > iters) gx = iters;
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL((k_hist_tiled<N>), dim3((unsigned)gx, (unsigned)groups, (unsigned)L),
                           dim3(kHistTiledThreads), lds, st, idx, (long)n_rows, (int)n_ch, (long)E, counts);
        VBQ_CHECK_LAUNCH("hist_tiled");
    }
    return VBQ_OK;
}

// ---------------------------------------------------------------------------- moments
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// one channel per blockIdx.y, contiguous elements
__global__ void __launch_bounds__(256)
k_moments_flat(const float *__restrict__ x, long n_per_ch, double *__restrict__ out, int vec_ok) {
    const int c = blockIdx.y;
    const float *src = x + (long)c * n_per_ch;
    double s1 = 0.0, s2 = 0.0;
    const long nq = vec_ok ? (n_per_ch >> 2) : 0;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
        const float4 v = *reinterpret_cast<const float4 *>(src + q * 4);
        const double a = v.x, b = v.y, cc = v.z, d = v.w;
        s1 += (a + b) + (cc + d);
        s2 += (a * a + b * b) + (cc * cc + d * d);
    }
    for (long i = nq * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_per_ch;
         i += (long)gridDim.x * blockDim.x) {
        const double a = src[i];
        s1 += a;
        s2 += a * a;
    }
    __shared__ double r1[4], r2[4];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { r1[w] = s1; r2[w] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&out[2 * c], (r1[0] + r1[1]) + (r1[2] + r1[3]));
        atomicAdd(&out[2 * c + 1], (r2[0] + r2[1]) + (r2[2] + r2[3]));
    }
}

// channel-last: thread t of a workgroup always sees channel (t % C) when blockDim % C == 0;
// generic C: each thread walks elements e = t0 + k*stride with stride a multiple of C.
__global__ void __launch_bounds__(256)
k_moments_bc(const float *__restrict__ x, long n_rows, int C, double *__restrict__ out) {
    extern __shared__ double acc[];   // [C][2]
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) acc[i] = 0.0;
    __syncthreads();
    const long E = n_rows * (long)C;
    const long nthreads = (long)gridDim.x * blockDim.x;
    const long stride = (nthreads / C) * C;     // <= nthreads (host guarantees nthreads >= C); a multiple of C
    const long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 < stride) {
        const int c = (int)(t0 % C);
        double s1 = 0.0, s2 = 0.0;
        for (long e = t0; e < E; e += stride) {
            const double a = x[e];
            s1 += a;
            s2 += a * a;
        }
        atomicAdd(&acc[2 * c], s1);
        atomicAdd(&acc[2 * c + 1], s2);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x)
        if (acc[i] != 0.0) atomicAdd(&out[i], acc[i]);
}

// ---------------------------------------------------------------------------- gather
__global__ void __launch_bounds__(256)
k_gather(const uint16_t *__restrict__ idx, long n_rows, int C, int layout, long E, int T,
         const float *__restrict__ tab, int per_lambda, float *__restrict__ out) {
    const int l = blockIdx.y;
    const uint16_t *src = idx + (long)l * E;
    float *dst = out + (long)l * E;
    const float *tl = tab + (per_lambda ? (long)l * C * T : 0);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        const int c = (C == 1) ? 0 : (layout == VBQ_LAYOUT_BC ? (int)(e % C) : (int)(e / n_rows));
        dst[e] = tl[(long)c * T + src[e]];
    }
}

}  // namespace
}  // namespace vbq

extern "C" int vbq_histogram_u16(const uint16_t *d_idx, int64_t n_rows, int32_t n_ch, int32_t layout,
                                 int32_t n_lambda, int32_t N, int64_t *d_counts, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows == 0 || (d_idx && d_counts), VBQ_ERR_INVALID_ARGUMENT, "vbq_histogram_u16: null pointer argument");
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_lambda >= 1 && n_lambda <= 65535 && n_ch <= 65535,
                VBQ_ERR_INVALID_ARGUMENT, "vbq_histogram_u16: bad sizes n_rows=%lld n_ch=%d n_lambda=%d",
                (long long)n_rows, n_ch, n_lambda);
    VBQ_REQUIRE(layout == VBQ_LAYOUT_BC || layout == VBQ_LAYOUT_CB, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_histogram_u16: unknown layout %d", layout);
    if (n_rows == 0) return VBQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(d_counts);
    switch (N) {
        case 10: return launch_hist<10>(d_idx, n_rows, n_ch, layout, n_lambda, cnt, st);
        case 8: return launch_hist<8>(d_idx, n_rows, n_ch, layout, n_lambda, cnt, st);
        case 6: return launch_hist<6>(d_idx, n_rows, n_ch, layout, n_lambda, cnt, st);
        case 4: return launch_hist<4>(d_idx, n_rows, n_ch, layout, n_lambda, cnt, st);
        default:
            set_error("vbq_histogram_u16: max_bits_per_coord N=%d not built (have 4, 6, 8, 10)", N);
            return VBQ_ERR_UNSUPPORTED;
    }
}

extern "C" int vbq_moments_f32(const float *d_x, int64_t n_rows, int32_t n_ch, int32_t layout, double *d_out,
                               void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows == 0 || (d_x && d_out), VBQ_ERR_INVALID_ARGUMENT, "vbq_moments_f32: null pointer argument");
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_ch <= 4096, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_moments_f32: bad sizes n_rows=%lld n_ch=%d (n_ch <= 4096)", (long long)n_rows, n_ch);
    VBQ_REQUIRE(layout == VBQ_LAYOUT_BC || layout == VBQ_LAYOUT_CB, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_moments_f32: unknown layout %d", layout);
    if (n_rows == 0) return VBQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t E = n_rows * (int64_t)n_ch;
    if (n_ch == 1 || layout == VBQ_LAYOUT_CB) {
        const int64_t n_per_ch = (n_ch == 1) ? E : n_rows;
        const int vec_ok = (reinterpret_cast<uintptr_t>(d_x) % 16 == 0) && (n_per_ch % 4 == 0 || n_ch == 1);
        int64_t gx = (n_per_ch / 4 + 255) / 256;
        const int64_t cap = 2048 / n_ch + 1;
        if (gx > cap) gx = cap;
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL(k_moments_flat, dim3((unsigned)gx, (unsigned)n_ch), dim3(256), 0, st, d_x, (long)n_per_ch,
                           d_out, vec_ok);
    } else {
        int64_t gx = (E + 256 * 16 - 1) / (256 * 16);
        if (gx > 2048) gx = 2048;
        if (gx < (n_ch + 255) / 256) gx = (n_ch + 255) / 256;
        hipLaunchKernelGGL(k_moments_bc, dim3((unsigned)gx), dim3(256), sizeof(double) * 2 * n_ch, st, d_x,
                           (long)n_rows, (int)n_ch, d_out);
    }
    VBQ_CHECK_LAUNCH("moments");
    return VBQ_OK;
}

extern "C" int vbq_gather_f32(const uint16_t *d_idx, int64_t n_rows, int32_t n_ch, int32_t layout,
                              int32_t n_lambda, int32_t N, const float *d_tab, int32_t tab_per_lambda,
                              float *d_out, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows == 0 || (d_idx && d_tab && d_out), VBQ_ERR_INVALID_ARGUMENT, "vbq_gather_f32: null pointer argument");
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_lambda >= 1 && n_lambda <= 65535 && N >= 0 && N <= 15,
                VBQ_ERR_INVALID_ARGUMENT, "vbq_gather_f32: bad sizes");
    VBQ_REQUIRE(layout == VBQ_LAYOUT_BC || layout == VBQ_LAYOUT_CB, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_gather_f32: unknown layout %d", layout);
    if (n_rows == 0) return VBQ_OK;
    const int64_t E = n_rows * (int64_t)n_ch;
    int64_t gx = (E + 255) / 256;
    const int64_t cap = 4096 / n_lambda + 1;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(k_gather, dim3((unsigned)gx, (unsigned)n_lambda), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), d_idx, (long)n_rows, (int)n_ch, (int)layout, (long)E,
                       table_size(N), d_tab, (int)tab_per_lambda, d_out);
    VBQ_CHECK_LAUNCH("gather");
    return VBQ_OK;
}

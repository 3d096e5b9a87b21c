// K2 histogram, K3 moments, rank-index table lookup.   gfx950 / CDNA4 only.
//
// K2 replaces the per-channel np.bincount of quantizer.py:104-105,138-140 and the Counter
// of ipynb:453.  One workgroup privatises the bins of one (lambda, channel[-group]) in LDS
// (u32), streams its share of the u16 indices with 16-B loads, and flushes the non-zero
// bins with one 64-bit global atomic each.  Integer adds: the result does not depend on
// the order of arrival, the grid shape or the number of GPUs.
#include <stdlib.h>

#include "vbq_common.h"

namespace vbq {
namespace {

#ifndef VBQ_HIST_THREADS
#define VBQ_HIST_THREADS 512
#endif
constexpr int kHistThreads = VBQ_HIST_THREADS;
#ifndef VBQ_HIST_U
#define VBQ_HIST_U 3
#endif

// LDS slot of rank index q.  In rank order every code point of bit levels 0..5 sits at
// q = 31 (mod 32) -- one LDS bank -- and those are exactly the bins that fill up at large
// lambda.  q ^ (q >> 6) is a bijection of [0, 2^11) that spreads every level evenly over the
// 32 banks (measured: the histogram pass went from 0.65 ms to the load-bound time).
// Indices that did not come from K1 may be anything up to 65535: the slot is masked into the 8192-word bin
// array (memory-safe, counts of such indices are meaningless -- vbq_index_max_u16 tells the caller beforehand).
template <int N>
__device__ __forceinline__ unsigned int bin_slot(unsigned int q) { return (q ^ (q >> 6)) & ((2u << N) - 1u); }

// Eight indices of one thread (one 16-B load) into the LDS histogram.
// An LDS atomic wave-instruction costs ~3 cycles per lane that shares a bank with another
// lane (same address included, unless ALL lanes agree), and at large lambda 50-90 % of the
// indices are one and the same bin.  So: (1) the thread counts how many of its eight indices
// equal its first one and issues the other seven adds only where they differ (few active
// lanes); (2) the lanes whose first index equals the wave leader's are summed with four
// ballots and added by one lane.  Spread-out distributions pay ~10 extra VALU ops per index.
// private copies of the bins per workgroup (lane & (K-1) picks one): 4 for up to 2048 bins (8 copies measured
// slower), 2 / 1 for the 4095 / 8191 bins of N = 11 / 12 -- always 32 KB of LDS
__host__ __device__ constexpr int hist_slots(int T) { return T + 1 <= 2048 ? 2048 : (T + 1 <= 4096 ? 4096 : 8192); }
__host__ __device__ constexpr int hist_copies(int T) { return 8192 / hist_slots(T); }
// The copies of a bin are adjacent words (word = 4 * slot + copy): lanes of different copies never meet on a
// bank, and zeroing / flushing the 32 KB moves 16 bytes per LDS instruction.

template <int N, int kHistCopies>
__device__ __forceinline__ void hist_add8(unsigned int *h, const uint4 v) {
    unsigned int s[8];
    {
        const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[2 * k] = bin_slot<N>(w[k] & 0xffffu);
            s[2 * k + 1] = bin_slot<N>(w[k] >> 16);
        }
    }
    unsigned int cnt = 1;
    bool eq[8];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        eq[k] = s[k] == s[0];
        cnt += eq[k] ? 1u : 0u;
    }
    const unsigned int lead = __builtin_amdgcn_readfirstlane(s[0]);
    const bool same = s[0] == lead;
    unsigned int tot = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        tot += (unsigned int)__popcll(__ballot(same && ((cnt >> b) & 1u))) << b;
    const unsigned long long same_mask = __ballot(same);
    const int first = __ffsll((long long)same_mask) - 1;
    const int lane = threadIdx.x & 63;
    unsigned int *hc = h + (lane & (kHistCopies - 1));                      // this lane's copy of the bins
    if (lane == first) atomicAdd(&hc[kHistCopies * lead], tot);
    if (!same) atomicAdd(&hc[kHistCopies * s[0]], cnt);
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (!eq[k]) atomicAdd(&hc[kHistCopies * s[k]], 1u);
}

// All indices of the workgroup belong to one channel: [l][c][n_per_ch] contiguous.
// assign_lut != nullptr (only with ONE workgroup per (lambda, channel), gridDim.x == 1): the workgroup owns its whole row of
// bins, so it STORES them (zeros included: no memset of the 67 MB array beforehand, no atomics) and writes the code-length
// model lut[count] next to them (quantizer.py:141-146 fused into the flush; models may be nullptr).
template <int N, typename CountT>
__global__ void __launch_bounds__(kHistThreads)
k_hist_flat(const uint16_t *__restrict__ idx, long n_per_ch, long ch_stride, int C, long E,
            CountT *__restrict__ counts, int vec_ok, int assign = 0, const float *__restrict__ assign_lut = nullptr,
            long lut_n = 0, float *__restrict__ models = nullptr) {
    constexpr int T = table_size(N);
    constexpr int kHistCopies = hist_copies(T);
    static_assert(T + 1 <= 8192, "bins are laid out for at most 8192 slots");
    __shared__ __align__(16) unsigned int h[8192];
    // assign == 2: the grid is (1, L, C) and walks the channels from the LAST one down, all lambdas of a channel together --
    // the order in which the solve kernel's output is most recent (and still in the memory-side cache) comes first
    const int c = assign == 2 ? C - 1 - (int)blockIdx.z : (int)blockIdx.y;
    const int l = assign == 2 ? (int)blockIdx.y : (int)blockIdx.z;
    // The 16-byte loads start at the first 16-byte boundary of this (lambda, channel) row: up to seven leading indices (odd
    // row counts shift every other row by two bytes) are counted one by one by the row's first workgroup, like the tail.
    const uint16_t *src0 = idx + (long)l * E + (long)c * ch_stride;
    const long head_max = (long)(((16u - (unsigned)(reinterpret_cast<uintptr_t>(src0) & 15u)) & 15u) >> 1);
    const long head = vec_ok ? (head_max < n_per_ch ? head_max : n_per_ch) : 0;
    const uint16_t *src = src0 + head;
    const long n_body = n_per_ch - head;
    const long noct = vec_ok ? (n_body >> 3) : 0;
    const long stride = (long)gridDim.x * blockDim.x;
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    // iterations in which every lane of the wave has two full loads take the aggregated path
    const int lane = threadIdx.x & 63;
    // Software-pipelined in two register stages of U 16-byte loads per lane: one stage is in flight while the
    // other is counted (Little's law: 20 waves per CU x 64 lanes x U x 16 B must cover the HBM latency); the
    // first stage is issued before the bins are cleared.  "full": every lane of the wave has all U loads.
    constexpr int U = VBQ_HIST_U;
    auto full = [&](long qq, int n) { return (qq - lane) + 63 + (long)(n - 1) * stride < noct; };
    // STAGES of U loads that every lane of this wave has (wave-uniform): stage s reads octets q0 + (s U + u) stride.  The loop
    // over them is written on that count, with NO condition around a load: with `if (haveB) load B` the compiler cannot tell
    // whether B is in flight, waits with vmcnt(1) / vmcnt(0) for A -- i.e. for B as well -- and the two register stages take
    // turns instead of overlapping (the same effect cost k_lookup_lds 10 %, EXPERIMENTS.md round 4).
    const long wq = q - lane;
    const long room = noct - 64 - wq - (long)(U - 1) * stride;
    const long stages = room >= 0 ? room / ((long)U * stride) + 1 : 0;
    uint4 A[U], B[U];
    auto load_stage = [&](uint4 (&R)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) R[u] = *reinterpret_cast<const uint4 *>(src + (q + u * stride) * 8);
        q += U * stride;
    };
    if (stages > 0) load_stage(A);
#pragma unroll
    for (int r = 0; r < 2048 / kHistThreads; ++r)                                                  // 32 KB
        reinterpret_cast<uint4 *>(h)[(int)threadIdx.x + r * kHistThreads] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (stages > 0) {
        const long pairs = (stages - 1) / 2;                    // iterations in which BOTH refills exist
        for (long p = 0; p < pairs; ++p) {
            load_stage(B);
#pragma unroll
            for (int u = 0; u < U; ++u) hist_add8<N, kHistCopies>(h, A[u]);
            load_stage(A);
#pragma unroll
            for (int u = 0; u < U; ++u) hist_add8<N, kHistCopies>(h, B[u]);
        }
        if (stages - (2 * pairs + 1) > 0) {                     // one more stage after the one A holds
            load_stage(B);
#pragma unroll
            for (int u = 0; u < U; ++u) hist_add8<N, kHistCopies>(h, A[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) hist_add8<N, kHistCopies>(h, B[u]);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) hist_add8<N, kHistCopies>(h, A[u]);
        }
    }
    for (; full(q, 1); q += stride) hist_add8<N, kHistCopies>(h, *reinterpret_cast<const uint4 *>(src + q * 8));
    for (; q < noct; q += stride) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + q * 8);
        const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicAdd(&h[kHistCopies * bin_slot<N>(w[k] & 0xffffu)], 1u);
            atomicAdd(&h[kHistCopies * bin_slot<N>(w[k] >> 16)], 1u);
        }
    }
    for (long i = noct * 8 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_body; i += stride)
        atomicAdd(&h[kHistCopies * bin_slot<N>(src[i])], 1u);
    if (blockIdx.x == 0 && (long)threadIdx.x < head) atomicAdd(&h[kHistCopies * bin_slot<N>(src0[threadIdx.x])], 1u);
    __syncthreads();
    CountT *dst = counts + ((long)l * C + c) * T;
    // Flush, the bins of a thread first read and summed, then written (and looked up) together: the LDS reads, the table
    // gathers and the stores of its FI bins overlap instead of following each other through per-bin branches.
    constexpr int FI = (T + kHistThreads - 1) / kHistThreads;
    unsigned int v[FI];
#pragma unroll
    for (int r = 0; r < FI; ++r) {
        const int i = (int)threadIdx.x + r * kHistThreads;
        const int ii = i < T ? i : T - 1;
        if constexpr (kHistCopies == 4) {
            const uint4 q4 = reinterpret_cast<const uint4 *>(h)[bin_slot<N>(ii)];
            v[r] = (q4.x + q4.y) + (q4.z + q4.w);
        } else if constexpr (kHistCopies == 2) {
            const uint2 q2 = reinterpret_cast<const uint2 *>(h)[bin_slot<N>(ii)];
            v[r] = q2.x + q2.y;
        } else {
            v[r] = h[bin_slot<N>(ii)];
        }
    }
    if (assign) {
#pragma unroll
        for (int r = 0; r < FI; ++r) {
            const int i = (int)threadIdx.x + r * kHistThreads;
            if (i < T) dst[i] = (CountT)v[r];
        }
        if (models) {
            float m[FI];
#pragma unroll
            for (int r = 0; r < FI; ++r) m[r] = assign_lut[(long)v[r] < lut_n ? (long)v[r] : lut_n - 1];
            float *mdst = models + ((long)l * C + c) * T;
#pragma unroll
            for (int r = 0; r < FI; ++r) {
                const int i = (int)threadIdx.x + r * kHistThreads;
                if (i < T) mdst[i] = m[r];
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < FI; ++r) {
            const int i = (int)threadIdx.x + r * kHistThreads;
            if (i < T && v[r]) atomicAdd(&dst[i], (CountT)v[r]);
        }
    }
}

// Channel-last [rows][C], C > 1: a workgroup owns 16 consecutive channels of one lambda.
constexpr int kHistTiledThreads = 1024;

template <int N, typename CountT>
__global__ void __launch_bounds__(kHistTiledThreads)
k_hist_tiled(const uint16_t *__restrict__ idx, long n_rows, int C, long E,
             CountT *__restrict__ counts) {
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
            atomicAdd(&hs[cl * TS + bin_slot<N>(src[r * C + c0 + cl])], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncg * T; i += blockDim.x) {
        const int ch = i / T, s = i - ch * T;
        const unsigned int v = hs[ch * TS + bin_slot<N>(s)];
        if (v) atomicAdd(&counts[((long)l * C + c0 + ch) * T + s], (CountT)v);
    }
}

// Rows [row_begin, row_end) of the full [L][n_rows x n_ch] index array (lambda planes E = n_rows * n_ch apart).
template <int N, typename CountT>
int launch_hist(const uint16_t *idx, int64_t n_rows, int32_t n_ch, int32_t layout, int32_t L,
                CountT *counts, int64_t row_begin, int64_t row_end, hipStream_t st) {
    const int64_t E = n_rows * (int64_t)n_ch;
    const int64_t n_sub = row_end - row_begin;
    const bool flat = (n_ch == 1) || (layout == VBQ_LAYOUT_CB);
    if (flat) {
        const uint16_t *src = idx + row_begin;
        const int64_t n_per_ch = n_sub;
        const int vec_ok = reinterpret_cast<uintptr_t>(src) % 2 == 0;       // every row finds its own 16-byte boundary (k_hist_flat)
        int64_t gx = (n_per_ch / 8 + kHistThreads - 1) / kHistThreads;
        int64_t cap = (int64_t)2048 / ((int64_t)n_ch * L) + 1;
        if (gx > cap) gx = cap;
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL((k_hist_flat<N, CountT>), dim3((unsigned)gx, (unsigned)n_ch, (unsigned)L), dim3(kHistThreads), 0, st,
                           src, (long)n_per_ch, (long)n_rows, (int)n_ch, (long)E, counts, vec_ok);
        VBQ_CHECK_LAUNCH("hist_flat");
    } else if constexpr (N > 10) {
        set_error("histogram: N=%d is built for channel-major planes only (VBQ_LAYOUT_CB, or n_ch = 1)", N);
        return VBQ_ERR_UNSUPPORTED;
    } else {
        constexpr int T = table_size(N);
        const size_t lds = sizeof(unsigned int) * kTileChannels * (T + 2);
        {   // per device and cheap: set on every call (a process may drive several GPUs)
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hist_tiled<N, CountT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
                return VBQ_ERR_LAUNCH;
            }
        }
        const int groups = (n_ch + kTileChannels - 1) / kTileChannels;
        int64_t gx = 512 / ((int64_t)groups * L) + 1;
        const int64_t iters = (n_sub + 63) / 64;
        if (gx > iters) gx = iters;
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL((k_hist_tiled<N, CountT>), dim3((unsigned)gx, (unsigned)groups, (unsigned)L),
                           dim3(kHistTiledThreads), lds, st, idx + row_begin * n_ch, (long)n_sub, (int)n_ch, (long)E, counts);
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
    // four 16-byte loads in flight per lane (one is not enough to cover the HBM latency at 8 waves per SIMD)
    const long stride = (long)gridDim.x * blockDim.x;
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; q + 3 * stride < nq; q += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(src + (q + u * stride) * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double a = v[u].x, b = v[u].y, cc = v[u].z, d = v[u].w;
            s1 += (a + b) + (cc + d);
            s2 += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    for (; q < nq; q += stride) {
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
// same layout in and out: one coalesced pass
__global__ void __launch_bounds__(256)
k_gather(const uint16_t *__restrict__ idx, long n_rows, int C, int layout, long E, int T,
         const float *__restrict__ tab, int per_lambda, float *__restrict__ out) {
    const int l = blockIdx.y;
    const uint16_t *src = idx + (long)l * E;
    float *dst = out + (long)l * E;
    const float *tl = tab + (per_lambda ? (long)l * C * T : 0);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        const int c = (C == 1) ? 0 : (layout == VBQ_LAYOUT_BC ? (int)(e % C) : (int)(e / n_rows));
        dst[e] = tl[(long)c * T + min((int)src[e], T - 1)];     // foreign indices >= T stay inside the table
    }
}

// ---------------------------------------------------------------------------- rate-distortion sums
// out[l] = { sum_e ((z - mu)^2 / (2 sigma^2)),  sum_e rate[l][c(e)][idx] }  with z = sorted table[c(e)][idx], everything
// accumulated in f64: the two terms of the Lagrangian the solve minimises, as a report (never used by a kernel).
constexpr int kRdChunk = 8;            // lambdas per workgroup: (mu, sigma) are read once per chunk, not once per lambda
__global__ void __launch_bounds__(256)
k_rd_sums(const float *__restrict__ mu, const float *__restrict__ sg, const uint16_t *__restrict__ idx, long n_rows, int C,
          int layout, long E, int T, const float *__restrict__ tab_sorted, const float *__restrict__ rate, int rate_per_lambda,
          int L, double *__restrict__ out) {
    const int l0 = blockIdx.y * kRdChunk;
    double sd[kRdChunk], sr[kRdChunk];
#pragma unroll
    for (int j = 0; j < kRdChunk; ++j) sd[j] = sr[j] = 0.0;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        const int c = (C == 1) ? 0 : (layout == VBQ_LAYOUT_BC ? (int)(e % C) : (int)(e / n_rows));
        const double m = (double)mu[e], s = (double)sg[e];
        const double w = 1.0 / (2.0 * s * s);
        const float *ts = tab_sorted + (long)c * T;
#pragma unroll
        for (int j = 0; j < kRdChunk; ++j) {
            const int l = l0 + j;
            if (l < L) {
                const int q = min((int)idx[(long)l * E + e], T - 1);
                const double d = (double)ts[q] - m;
                sd[j] += d * d * w;
                if (rate) sr[j] += (double)rate[(rate_per_lambda ? (long)l * C * T : 0) + (long)c * T + q];
            }
        }
    }
    __shared__ double part[2][kRdChunk][4];
    const int w4 = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < kRdChunk; ++j) {
        double a = sd[j], b = sr[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o);
            b += __shfl_down(b, o);
        }
        if ((threadIdx.x & 63) == 0) { part[0][j][w4] = a; part[1][j][w4] = b; }
    }
    __syncthreads();
    if (threadIdx.x < kRdChunk && l0 + (int)threadIdx.x < L) {
        const int j = threadIdx.x;
        atomicAdd(&out[2 * (l0 + j)], part[0][j][0] + part[0][j][1] + part[0][j][2] + part[0][j][3]);
        atomicAdd(&out[2 * (l0 + j) + 1], part[1][j][0] + part[1][j][1] + part[1][j][2] + part[1][j][3]);
    }
}

// idx and out in different layouts: 32 x 32 tiles through LDS, both sides coalesced.
// in_rows x in_cols is the shape of the idx matrix as stored ([C][B] for CB, [B][C] for BC).
__global__ void __launch_bounds__(256)
k_gather_transpose(const uint16_t *__restrict__ idx, long in_rows, long in_cols, int in_layout, long E, int C, int T,
                   const float *__restrict__ tab, int per_lambda, float *__restrict__ out) {
    __shared__ float tile[32][33];
    const int l = blockIdx.z;
    const uint16_t *src = idx + (long)l * E;
    float *dst = out + (long)l * E;
    const float *tl = tab + (per_lambda ? (long)l * C * T : 0);
    const long ctiles = (in_cols + 31) / 32;                         // tiles numbered along blockIdx.x, columns fastest
    const long r0 = ((long)blockIdx.x / ctiles) * 32, c0 = ((long)blockIdx.x % ctiles) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < in_rows && c < in_cols) {
            const int ch = in_layout == VBQ_LAYOUT_CB ? (int)r : (int)c;
            tile[ty + 8 * k][tx] = tl[(long)ch * T + min((int)src[r * in_cols + c], T - 1)];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long c = c0 + ty + 8 * k, r = r0 + tx;                 // output row = input column
        if (r < in_rows && c < in_cols) dst[c * in_rows + r] = tile[tx][ty + 8 * k];
    }
}

__global__ void __launch_bounds__(256)
k_transpose(const float *__restrict__ in, long rows, long cols, float *__restrict__ out) {
    __shared__ float tile[64][65];
    const long ctiles = (cols + 63) / 64;        // tiles numbered along blockIdx.x (columns fastest): no 65535-tile limit on rows
    const long r0 = ((long)blockIdx.x / ctiles) * 64, c0 = ((long)blockIdx.x % ctiles) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const long r = r0 + ty + 4 * k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 4 * k][tx] = in[r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const long c = c0 + ty + 4 * k, r = r0 + tx;
        if (r < rows && c < cols) out[c * rows + r] = tile[tx][ty + 4 * k];
    }
}

// Same tiles with 16-byte global accesses (rows and cols multiples of 4, 16-byte aligned pointers).
__global__ void __launch_bounds__(256)
k_transpose_v4(const float *__restrict__ in, long rows, long cols, float *__restrict__ out) {
    __shared__ float tile[64][65];
    const long ctiles = (cols + 63) / 64;
    const long r0 = ((long)blockIdx.x / ctiles) * 64, c0 = ((long)blockIdx.x % ctiles) * 64;
    if (r0 + 64 <= rows && c0 + 64 <= cols) {
        // whole tile (workgroup-uniform): the four loads of a thread go out together, then the LDS writes -- with the bounds
        // test around every access each load was waited for before the next one was issued
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k, lr = i >> 4, lc = (i & 15) * 4;
            v[k] = *reinterpret_cast<const float4 *>(in + (r0 + lr) * cols + c0 + lc);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k, lr = i >> 4, lc = (i & 15) * 4;
            tile[lr][lc] = v[k].x; tile[lr][lc + 1] = v[k].y; tile[lr][lc + 2] = v[k].z; tile[lr][lc + 3] = v[k].w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k, lc = i >> 4, lr = (i & 15) * 4;
            v[k] = make_float4(tile[lr][lc], tile[lr + 1][lc], tile[lr + 2][lc], tile[lr + 3][lc]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k, lc = i >> 4, lr = (i & 15) * 4;
            *reinterpret_cast<float4 *>(out + (c0 + lc) * rows + r0 + lr) = v[k];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + 256 * k, lr = i >> 4, lc = (i & 15) * 4;
        const long r = r0 + lr, c = c0 + lc;
        if (r < rows && c < cols) {
            const float4 v = *reinterpret_cast<const float4 *>(in + r * cols + c);
            tile[lr][lc] = v.x; tile[lr][lc + 1] = v.y; tile[lr][lc + 2] = v.z; tile[lr][lc + 3] = v.w;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + 256 * k, lc = i >> 4, lr = (i & 15) * 4;
        const long c = c0 + lc, r = r0 + lr;
        if (r < rows && c < cols)
            *reinterpret_cast<float4 *>(out + c * rows + r) = make_float4(tile[lr][lc], tile[lr + 1][lc], tile[lr + 2][lc], tile[lr + 3][lc]);
    }
}

// Batched transpose of 2- or 4-byte elements: in [batch][rows][cols] -> out [batch][cols][rows], 64 x 64 tiles through LDS
// with 16-byte global accesses on both sides (rows and cols multiples of the vector width, 16-byte aligned planes).
// The u16 form turns the plane indices [L][C][B] of the fast kernels back into the caller's channel-last [L][B][C].
template <typename T>
__global__ void __launch_bounds__(256)
k_transpose_batched_vec(const T *__restrict__ in, long rows, long cols, T *__restrict__ out) {
    constexpr int V = 16 / sizeof(T);                        // elements per 16-byte access
    constexpr int VPR = 64 / V;                              // vectors per tile row
    constexpr int PER = 64 * VPR / 256;                      // vectors per thread
    struct alignas(16) Vec { T e[V]; };
    __shared__ T tile[64][64 + 2];
    in += (long)blockIdx.y * rows * cols;
    out += (long)blockIdx.y * rows * cols;
    const long ctiles = (cols + 63) / 64;
    const long r0 = ((long)blockIdx.x / ctiles) * 64, c0 = ((long)blockIdx.x % ctiles) * 64;
    if (r0 + 64 <= rows && c0 + 64 <= cols) {                  // whole tile: loads together, then LDS (see k_transpose_v4)
        Vec v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + 256 * k, lr = i / VPR, lc = (i % VPR) * V;
            v[k] = *reinterpret_cast<const Vec *>(in + (r0 + lr) * cols + c0 + lc);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + 256 * k, lr = i / VPR, lc = (i % VPR) * V;
#pragma unroll
            for (int e = 0; e < V; ++e) tile[lr][lc + e] = v[k].e[e];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + 256 * k, lc = i / VPR, lr = (i % VPR) * V;
#pragma unroll
            for (int e = 0; e < V; ++e) v[k].e[e] = tile[lr + e][lc];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + 256 * k, lc = i / VPR, lr = (i % VPR) * V;
            *reinterpret_cast<Vec *>(out + (c0 + lc) * rows + r0 + lr) = v[k];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = threadIdx.x + 256 * k, lr = i / VPR, lc = (i % VPR) * V;
        const long r = r0 + lr, c = c0 + lc;
        if (r < rows && c < cols) {
            const Vec v = *reinterpret_cast<const Vec *>(in + r * cols + c);
#pragma unroll
            for (int e = 0; e < V; ++e) tile[lr][lc + e] = v.e[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = threadIdx.x + 256 * k, lc = i / VPR, lr = (i % VPR) * V;
        const long c = c0 + lc, r = r0 + lr;
        if (r < rows && c < cols) {
            Vec v;
#pragma unroll
            for (int e = 0; e < V; ++e) v.e[e] = tile[lr + e][lc];
            *reinterpret_cast<Vec *>(out + c * rows + r) = v;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_transpose_batched(const T *__restrict__ in, long rows, long cols, T *__restrict__ out) {
    __shared__ T tile[64][65];
    in += (long)blockIdx.y * rows * cols;
    out += (long)blockIdx.y * rows * cols;
    const long ctiles = (cols + 63) / 64;
    const long r0 = ((long)blockIdx.x / ctiles) * 64, c0 = ((long)blockIdx.x % ctiles) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const long r = r0 + ty + 4 * k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 4 * k][tx] = in[r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const long c = c0 + ty + 4 * k, r = r0 + tx;
        if (r < rows && c < cols) out[c * rows + r] = tile[tx][ty + 4 * k];
    }
}

template <typename T>
int launch_transpose_batched(const T *in, int64_t batch, int64_t rows, int64_t cols, T *out, hipStream_t st) {
    constexpr int V = 16 / sizeof(T);
    const bool vec = rows % V == 0 && cols % V == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 &&
                     (rows * cols * (int64_t)sizeof(T)) % 16 == 0;
    const dim3 grid((unsigned)(((cols + 63) / 64) * ((rows + 63) / 64)), (unsigned)batch);      // tiles on x, planes on y
    if (vec)
        hipLaunchKernelGGL((k_transpose_batched_vec<T>), grid, dim3(256), 0, st, in, (long)rows, (long)cols, out);
    else
        hipLaunchKernelGGL((k_transpose_batched<T>), grid, dim3(256), 0, st, in, (long)rows, (long)cols, out);
    VBQ_CHECK_LAUNCH("transpose_planes");
    return VBQ_OK;
}

}  // namespace
}  // namespace vbq

namespace vbq {
namespace {
template <typename CountT>
int histogram_entry(const char *who, const uint16_t *d_idx, int64_t n_rows, int32_t n_ch, int32_t layout,
                    int32_t n_lambda, int32_t N, CountT *cnt, int64_t row_begin, int64_t row_end, void *stream) {
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_lambda >= 1 && n_lambda <= 65535 && n_ch <= 65535,
                VBQ_ERR_INVALID_ARGUMENT, "%s: bad sizes n_rows=%lld n_ch=%d n_lambda=%d", who, (long long)n_rows, n_ch,
                n_lambda);
    VBQ_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= n_rows, VBQ_ERR_INVALID_ARGUMENT,
                "%s: row range [%lld, %lld) outside [0, %lld)", who, (long long)row_begin, (long long)row_end, (long long)n_rows);
    VBQ_REQUIRE(row_begin == row_end || (d_idx && cnt), VBQ_ERR_INVALID_ARGUMENT, "%s: null pointer argument", who);
    VBQ_REQUIRE(layout == VBQ_LAYOUT_BC || layout == VBQ_LAYOUT_CB, VBQ_ERR_INVALID_ARGUMENT, "%s: unknown layout %d",
                who, layout);
    VBQ_REQUIRE(sizeof(CountT) == 8 || n_rows <= 0x7fffffffLL, VBQ_ERR_INVALID_ARGUMENT,
                "%s: %lld rows per channel can overflow 32-bit counters", who, (long long)n_rows);
    if (row_begin == row_end) return VBQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define VBQ_DISPATCH_N(NN) \
    case NN: return launch_hist<NN, CountT>(d_idx, n_rows, n_ch, layout, n_lambda, cnt, row_begin, row_end, st);
    switch (N) {
        VBQ_FOR_EACH_N(VBQ_DISPATCH_N)
        default:
            set_error("%s: max_bits_per_coord N=%d not built (have 4 ... 12; only 10 with VBQ_ONLY_N10)", who, N);
            return VBQ_ERR_UNSUPPORTED;
    }
#undef VBQ_DISPATCH_N
}

// out[i] = (period ? i % period : 0) + lut[min(counts[i], lut_n - 1)]
template <typename CountT>
__global__ void __launch_bounds__(256)
k_lut_lengths(const CountT *__restrict__ counts, long n, const float *__restrict__ lut, long lut_n, int period,
              float *__restrict__ out, float *__restrict__ out_model) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        long k = (long)counts[i];
        k = k < 0 ? 0 : (k >= lut_n ? lut_n - 1 : k);
        const float m = lut[k];
        if (out_model) out_model[i] = m;
        if (out) out[i] = period ? __fadd_rn((float)(int)(i % period), m) : m;
    }
}
// period == 0, model output only, n % 4 == 0, 16-byte aligned: 4 entries per thread and iteration (the [L][C][T] model table
// of a Kodak-size build is 16.7 M entries: 67 + 67 MB through this kernel)
template <typename CountT>
__global__ void __launch_bounds__(256)
k_lut_models_v4(const CountT *__restrict__ counts, long n4, const float *__restrict__ lut, long lut_n, float *__restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        long k[4];
        if constexpr (sizeof(CountT) == 4) {
            const int4 c = reinterpret_cast<const int4 *>(counts)[i];
            k[0] = c.x; k[1] = c.y; k[2] = c.z; k[3] = c.w;
        } else {
            const longlong2 a = reinterpret_cast<const longlong2 *>(counts)[2 * i], b = reinterpret_cast<const longlong2 *>(counts)[2 * i + 1];
            k[0] = a.x; k[1] = a.y; k[2] = b.x; k[3] = b.y;
        }
        float m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long kk = k[j] < 0 ? 0 : (k[j] >= lut_n ? lut_n - 1 : k[j]);
            m[j] = lut[kk];
        }
        reinterpret_cast<float4 *>(out)[i] = make_float4(m[0], m[1], m[2], m[3]);
    }
}
}  // namespace
}  // namespace vbq

extern "C" int vbq_histogram_u16(const uint16_t *d_idx, int64_t n_rows, int32_t n_ch, int32_t layout,
                                 int32_t n_lambda, int32_t N, int64_t *d_counts, void *stream) {
    return vbq::histogram_entry<unsigned long long>("vbq_histogram_u16", d_idx, n_rows, n_ch, layout, n_lambda, N,
                                                    reinterpret_cast<unsigned long long *>(d_counts), 0, n_rows, stream);
}

extern "C" int vbq_histogram_u16_i32(const uint16_t *d_idx, int64_t n_rows, int32_t n_ch, int32_t layout,
                                     int32_t n_lambda, int32_t N, int32_t *d_counts, void *stream) {
    return vbq::histogram_entry<unsigned int>("vbq_histogram_u16_i32", d_idx, n_rows, n_ch, layout, n_lambda, N,
                                              reinterpret_cast<unsigned int *>(d_counts), 0, n_rows, stream);
}

extern "C" int vbq_histogram_rows_u16(const uint16_t *d_idx, int64_t n_rows, int32_t n_ch, int32_t layout,
                                      int32_t n_lambda, int32_t N, void *d_counts, int32_t counts_are_i32,
                                      int64_t row_begin, int64_t row_end, void *stream) {
    if (counts_are_i32)
        return vbq::histogram_entry<unsigned int>("vbq_histogram_rows_u16", d_idx, n_rows, n_ch, layout, n_lambda, N,
                                                  reinterpret_cast<unsigned int *>(d_counts), row_begin, row_end, stream);
    return vbq::histogram_entry<unsigned long long>("vbq_histogram_rows_u16", d_idx, n_rows, n_ch, layout, n_lambda, N,
                                                    reinterpret_cast<unsigned long long *>(d_counts), row_begin, row_end, stream);
}

namespace vbq {
namespace {
template <int N, typename CountT>
int launch_hist_assign(const uint16_t *idx, int64_t n_rows, int32_t n_ch, int32_t L, CountT *counts, const float *lut,
                       int64_t lut_n, float *models, hipStream_t st) {
    const int64_t E = n_rows * (int64_t)n_ch;
    const int vec_ok = reinterpret_cast<uintptr_t>(idx) % 2 == 0;           // every row finds its own 16-byte boundary (k_hist_flat)
    // last channel first (measured on the Kodak-24 build of round 2, whose K1 walked the channels in order: 0.7745 against
    // 0.7793 ms per step, three runs each; K1's resident grid of round 3 writes all channels side by side, so what is still in
    // the memory-side cache is the tail of EVERY row and the order no longer matters)
    if (n_ch <= 65535)
        hipLaunchKernelGGL((k_hist_flat<N, CountT>), dim3(1u, (unsigned)L, (unsigned)n_ch), dim3(kHistThreads), 0, st, idx, (long)n_rows,
                           (long)n_rows, (int)n_ch, (long)E, counts, vec_ok, 2, lut, (long)lut_n, models);
    else
        hipLaunchKernelGGL((k_hist_flat<N, CountT>), dim3(1u, (unsigned)n_ch, (unsigned)L), dim3(kHistThreads), 0, st, idx, (long)n_rows,
                           (long)n_rows, (int)n_ch, (long)E, counts, vec_ok, 1, lut, (long)lut_n, models);
    VBQ_CHECK_LAUNCH("hist_assign");
    return VBQ_OK;
}
}  // namespace
}  // namespace vbq

extern "C" int vbq_histogram_models_u16(const uint16_t *d_idx, int64_t n_rows, int32_t n_ch, int32_t n_lambda, int32_t N,
                                        void *d_counts, int32_t counts_are_i32, const float *d_lut, int64_t lut_n,
                                        float *d_models, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_lambda >= 1 && n_lambda <= 65535 && n_ch <= 65535, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_histogram_models_u16: bad sizes n_rows=%lld n_ch=%d n_lambda=%d", (long long)n_rows, n_ch, n_lambda);
    VBQ_REQUIRE(d_idx && d_counts && (!d_models || (d_lut && lut_n >= 1)), VBQ_ERR_INVALID_ARGUMENT,
                "vbq_histogram_models_u16: null pointer argument");
    VBQ_REQUIRE(!counts_are_i32 || n_rows <= 0x7fffffffLL, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_histogram_models_u16: %lld rows per channel can overflow 32-bit counters", (long long)n_rows);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t n_bins = (int64_t)n_lambda * n_ch * table_size(N);
    // one workgroup per (lambda, channel) is the launch shape of the plain histogram whenever there are at least 2048 rows of
    // bins (cap = 2048 / (C L) + 1 = 1): then the fused form applies; otherwise compose it from the plain entry points
    const bool fused = (int64_t)n_ch * n_lambda >= 2048 && N >= 4 && N <= 12 && n_rows > 0;
    if (fused) {
#define VBQ_DISPATCH_N(NN)                                                                                                   \
    case NN:                                                                                                                 \
        return counts_are_i32 ? launch_hist_assign<NN, unsigned int>(d_idx, n_rows, n_ch, n_lambda,                          \
                                                                     reinterpret_cast<unsigned int *>(d_counts), d_lut, lut_n, \
                                                                     d_models, st)                                           \
                              : launch_hist_assign<NN, unsigned long long>(d_idx, n_rows, n_ch, n_lambda,                    \
                                                                           reinterpret_cast<unsigned long long *>(d_counts), \
                                                                           d_lut, lut_n, d_models, st);
        switch (N) {
            VBQ_FOR_EACH_N(VBQ_DISPATCH_N)
            default: break;
        }
#undef VBQ_DISPATCH_N
        set_error("vbq_histogram_models_u16: max_bits_per_coord N=%d not built", N);
        return VBQ_ERR_UNSUPPORTED;
    }
    if (hipMemsetAsync(d_counts, 0, (size_t)n_bins * (counts_are_i32 ? 4 : 8), st) != hipSuccess) {
        set_error("vbq_histogram_models_u16: hipMemsetAsync failed");
        return VBQ_ERR_LAUNCH;
    }
    int r = vbq_histogram_rows_u16(d_idx, n_rows, n_ch, VBQ_LAYOUT_CB, n_lambda, N, d_counts, counts_are_i32, 0, n_rows, stream);
    if (r != VBQ_OK || !d_models) return r;
    return vbq_code_lengths_from_counts(d_counts, counts_are_i32, n_bins, d_lut, lut_n, 0, nullptr, d_models, stream);
}

extern "C" int vbq_code_lengths_from_counts(const void *d_counts, int32_t counts_are_i32, int64_t n,
                                            const float *d_lut, int64_t lut_n, int32_t level_period,
                                            float *d_out_len, float *d_out_model, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n >= 0 && lut_n >= 1 && level_period >= 0, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_code_lengths_from_counts: bad sizes n=%lld lut_n=%lld period=%d", (long long)n, (long long)lut_n, level_period);
    if (n == 0) return VBQ_OK;
    VBQ_REQUIRE(d_counts && d_lut && (d_out_len || d_out_model), VBQ_ERR_INVALID_ARGUMENT,
                "vbq_code_lengths_from_counts: null pointer argument");
    int64_t gx = (n + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (level_period == 0 && !d_out_len && n % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(d_counts) | reinterpret_cast<uintptr_t>(d_out_model)) & 15) == 0) {
        int64_t g4 = (n / 4 + 255) / 256;
        if (g4 > 8192) g4 = 8192;
        if (counts_are_i32)
            hipLaunchKernelGGL(k_lut_models_v4<int32_t>, dim3((unsigned)g4), dim3(256), 0, st, reinterpret_cast<const int32_t *>(d_counts),
                               (long)(n / 4), d_lut, (long)lut_n, d_out_model);
        else
            hipLaunchKernelGGL(k_lut_models_v4<long long>, dim3((unsigned)g4), dim3(256), 0, st,
                               reinterpret_cast<const long long *>(d_counts), (long)(n / 4), d_lut, (long)lut_n, d_out_model);
        VBQ_CHECK_LAUNCH("code_lengths_from_counts");
        return VBQ_OK;
    }
    if (counts_are_i32)
        hipLaunchKernelGGL(k_lut_lengths<int32_t>, dim3((unsigned)gx), dim3(256), 0, st, reinterpret_cast<const int32_t *>(d_counts),
                           (long)n, d_lut, (long)lut_n, (int)level_period, d_out_len, d_out_model);
    else
        hipLaunchKernelGGL(k_lut_lengths<long long>, dim3((unsigned)gx), dim3(256), 0, st,
                           reinterpret_cast<const long long *>(d_counts), (long)n, d_lut, (long)lut_n, (int)level_period, d_out_len,
                           d_out_model);
    VBQ_CHECK_LAUNCH("code_lengths_from_counts");
    return VBQ_OK;
}

// ---------------------------------------------------------------------------- packed counters for the all-reduce
// Three 21-bit counters per int64 word: an integer SUM all-reduce of the words adds the three fields
// independently as long as every global count stays below 2^21 (no carry between fields), and moves
// 2.67 instead of 4 bytes per bin over xGMI.  n bins -> ceil(n / 3) words; missing fields are zero.
namespace vbq {
namespace {
__global__ void __launch_bounds__(256)
k_pack3x21(const int32_t *__restrict__ c, long n, long nw, unsigned long long *__restrict__ w, unsigned int limit,
           unsigned int *__restrict__ overflow) {
    bool over = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (long)gridDim.x * blockDim.x) {
        const long b = 3 * i;
        const unsigned int f0 = (unsigned int)c[b];
        const unsigned int f1 = b + 1 < n ? (unsigned int)c[b + 1] : 0u;
        const unsigned int f2 = b + 2 < n ? (unsigned int)c[b + 2] : 0u;
        over = over || f0 >= limit || f1 >= limit || f2 >= limit;        // negative counts are huge as unsigned
        w[i] = (unsigned long long)(f0 & 0x1fffffu) | ((unsigned long long)(f1 & 0x1fffffu) << 21) |
               ((unsigned long long)(f2 & 0x1fffffu) << 42);
    }
    if (overflow && __ballot(over) != 0ull && (threadIdx.x & 63) == 0) atomicOr(overflow, 1u);
}
__global__ void __launch_bounds__(256)
k_unpack3x21(const unsigned long long *__restrict__ w, long n, long nw, int32_t *__restrict__ c) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (long)gridDim.x * blockDim.x) {
        const unsigned long long v = w[i];
        const long b = 3 * i;
        c[b] = (int32_t)(v & 0x1fffffu);
        if (b + 1 < n) c[b + 1] = (int32_t)((v >> 21) & 0x1fffffu);
        if (b + 2 < n) c[b + 2] = (int32_t)((v >> 42) & 0x1fffffu);
    }
}
}  // namespace
}  // namespace vbq

extern "C" int vbq_pack_counts_3x21(const int32_t *d_counts, int64_t n, int64_t *d_words, int32_t n_ranks,
                                    uint32_t *d_overflow, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n >= 0, VBQ_ERR_INVALID_ARGUMENT, "vbq_pack_counts_3x21: n < 0");
    VBQ_REQUIRE(n_ranks >= 1 && n_ranks <= (1 << 20), VBQ_ERR_INVALID_ARGUMENT, "vbq_pack_counts_3x21: n_ranks=%d", n_ranks);
    if (n == 0) return VBQ_OK;
    VBQ_REQUIRE(d_counts && d_words, VBQ_ERR_INVALID_ARGUMENT, "vbq_pack_counts_3x21: null pointer");
    const int64_t nw = (n + 2) / 3;
    int64_t gx = (nw + 255) / 256;
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(k_pack3x21, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_counts, (long)n, (long)nw,
                       reinterpret_cast<unsigned long long *>(d_words), (unsigned int)((1u << 21) / (unsigned)n_ranks), d_overflow);
    VBQ_CHECK_LAUNCH("pack_counts");
    return VBQ_OK;
}

extern "C" int vbq_unpack_counts_3x21(const int64_t *d_words, int64_t n, int32_t *d_counts, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n >= 0, VBQ_ERR_INVALID_ARGUMENT, "vbq_unpack_counts_3x21: n < 0");
    if (n == 0) return VBQ_OK;
    VBQ_REQUIRE(d_counts && d_words, VBQ_ERR_INVALID_ARGUMENT, "vbq_unpack_counts_3x21: null pointer");
    const int64_t nw = (n + 2) / 3;
    int64_t gx = (nw + 255) / 256;
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(k_unpack3x21, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned long long *>(d_words), (long)n, (long)nw, d_counts);
    VBQ_CHECK_LAUNCH("unpack_counts");
    return VBQ_OK;
}

namespace vbq {
namespace {
__global__ void __launch_bounds__(256)
k_index_max(const uint16_t *__restrict__ idx, long n, unsigned int *__restrict__ out) {
    unsigned int m = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = max(m, (unsigned int)idx[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned int)__shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
}  // namespace
}  // namespace vbq

namespace vbq {
namespace {
__global__ void __launch_bounds__(256)
k_check_inputs(const float *__restrict__ mu, const float *__restrict__ sg, long n, unsigned int *__restrict__ out) {
    unsigned int bad_mu = 0, bad_sg = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float m = mu[i], s = sg[i];
        bad_mu += !(fabsf(m) <= 3.4028234e38f) ? 1u : 0u;                       // NaN or infinity
        bad_sg += !(s > 0.0f && s <= 3.4028234e38f) ? 1u : 0u;                  // NaN, infinity, zero or negative
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { bad_mu += __shfl_down(bad_mu, o, 64); bad_sg += __shfl_down(bad_sg, o, 64); }
    if ((threadIdx.x & 63) == 0 && (bad_mu | bad_sg)) { atomicAdd(&out[0], bad_mu); atomicAdd(&out[1], bad_sg); }
}
}  // namespace
}  // namespace vbq

extern "C" int vbq_check_inputs_f32(const float *d_mu, const float *d_sigma, int64_t n, uint32_t *d_bad, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n >= 0, VBQ_ERR_INVALID_ARGUMENT, "vbq_check_inputs_f32: n < 0");
    if (n == 0) return VBQ_OK;
    VBQ_REQUIRE(d_mu && d_sigma && d_bad, VBQ_ERR_INVALID_ARGUMENT, "vbq_check_inputs_f32: null pointer");
    int64_t gx = (n + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_check_inputs, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_mu, d_sigma, (long)n, d_bad);
    VBQ_CHECK_LAUNCH("check_inputs");
    return VBQ_OK;
}

extern "C" int vbq_index_max_u16(const uint16_t *d_idx, int64_t n, uint32_t *d_max, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n >= 0, VBQ_ERR_INVALID_ARGUMENT, "vbq_index_max_u16: n < 0");
    if (n == 0) return VBQ_OK;
    VBQ_REQUIRE(d_idx && d_max, VBQ_ERR_INVALID_ARGUMENT, "vbq_index_max_u16: null pointer");
    int64_t gx = (n + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_index_max, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_idx, (long)n, d_max);
    VBQ_CHECK_LAUNCH("index_max");
    return VBQ_OK;
}

// ---------------------------------------------------------------------------- the notebook's moment, in NumPy's order
// empirical_std = np.sqrt(np.mean(vecs_u.ravel()**2)) (ipynb:374) is a float32 reduction, and float32 sums depend on
// their order.  NumPy's order: the reduce loop hands its pairwise routine blocks of 8192 elements (the iterator's
// buffer) and accumulates the block results one after the other; inside a block, halves are split on multiples of 8
// down to runs of <= 128 elements, which are summed with 8 interleaved accumulators combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) (loops_utils.h.src, *_pairwise_sum).  Reproduced here operation for operation:
// a full block is 64 runs of 128 -> one workgroup; the ragged last block and the chain over the block sums run on one
// thread out of LDS.  oracle/vbq_oracle.c restates the same order in C and is pinned against np.sum.
namespace vbq {
namespace {
constexpr int kNpBlock = 8192;
constexpr int kNpRunStride = 136;      // 128 + 8 words: the 64 runs of a block start on different LDS banks

// SQ: sum of squares (the notebook's moment), else the plain sum (np.sum of the per-image code lengths, utils.py:547-552).
// blockIdx.y: the row of a [rows][n] batch (row sums are independent reductions of n contiguous elements each).
template <bool SQ>
__global__ void __launch_bounds__(256)
k_np_block_sums(const float *__restrict__ x, long n, long nfull, float *__restrict__ block_sums, int vec) {
    __shared__ float sq[64 * kNpRunStride];
    __shared__ float racc[512];
    __shared__ float leaf[64];
    const float *src = x + (long)blockIdx.y * n + (long)blockIdx.x * kNpBlock;
    block_sums += (long)blockIdx.y * nfull;
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = (k * 256 + t) * 4;                      // element within the block; 4 | 128: one run per float4
        float4 v;
        if (vec) v = *reinterpret_cast<const float4 *>(src + e);
        else v = make_float4(src[e], src[e + 1], src[e + 2], src[e + 3]);      // rows that do not start on 16 bytes
        float *d = sq + (e >> 7) * kNpRunStride + (e & 127);
        if (SQ) { d[0] = __fmul_rn(v.x, v.x); d[1] = __fmul_rn(v.y, v.y); d[2] = __fmul_rn(v.z, v.z); d[3] = __fmul_rn(v.w, v.w); }
        else { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int p = t + 256 * h, run = p >> 3, j = p & 7;
        const float *a = sq + run * kNpRunStride + j;
        float r = a[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) r = __fadd_rn(r, a[8 * i]);
        racc[p] = r;
    }
    __syncthreads();
    if (t < 64) {
        const float *r = racc + 8 * t;
        leaf[t] = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                            __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
    }
    __syncthreads();
    for (int s2 = 1; s2 < 64; s2 <<= 1) {
        if (t < 64 && (t & (2 * s2 - 1)) == 0) leaf[t] = __fadd_rn(leaf[t], leaf[t + s2]);
        __syncthreads();
    }
    if (t == 0) block_sums[blockIdx.x] = leaf[0];
}

// pairwise sum of n <= 8192 ready-made squares (one thread; recursion depth <= 7)
__device__ __noinline__ float np_pairwise(const float *a, int n) {
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; ++i) res = __fadd_rn(res, a[i]);
        return res;
    }
    if (n <= 128) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], a[i + j]);
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                              __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, a[i]);
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    const float lo = np_pairwise(a, n2);
    return __fadd_rn(lo, np_pairwise(a + n2, n - n2));
}

template <bool SQ>
__global__ void __launch_bounds__(256)
k_np_finish(const float *__restrict__ x, long n, const float *__restrict__ block_sums, float *__restrict__ out) {
    __shared__ float buf[kNpBlock];
    const long nfull = n / kNpBlock;
    x += (long)blockIdx.x * n;
    block_sums += (long)blockIdx.x * nfull;
    float acc = 0.0f;
    for (long b0 = 0; b0 < nfull; b0 += kNpBlock) {           // the chain over the block sums, 8192 at a time out of LDS
        const int m = (int)(nfull - b0 < kNpBlock ? nfull - b0 : kNpBlock);
        for (int i = threadIdx.x; i < m; i += blockDim.x) buf[i] = block_sums[b0 + i];
        __syncthreads();
        if (threadIdx.x == 0)
            for (int i = 0; i < m; ++i) acc = __fadd_rn(acc, buf[i]);
        __syncthreads();
    }
    const int tail = (int)(n - nfull * kNpBlock);
    if (tail > 0) {
        for (int i = threadIdx.x; i < tail; i += blockDim.x) {
            const float v = x[nfull * kNpBlock + i];
            buf[i] = SQ ? __fmul_rn(v, v) : v;
        }
        __syncthreads();
        if (threadIdx.x == 0) acc = __fadd_rn(acc, np_pairwise(buf, tail));
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

template <bool SQ>
int launch_np_sums(const float *d_x, int64_t n, int64_t rows, float *d_out, float *bs, hipStream_t st) {
    const int64_t nfull = n / kNpBlock;
    // a row of the batch starts on 16 bytes when the base does and 4 | n; otherwise 4-byte loads
    const int vec = (reinterpret_cast<uintptr_t>(d_x) & 15) == 0 && (rows == 1 || n % 4 == 0);
    if (nfull > 0) {
        hipLaunchKernelGGL((k_np_block_sums<SQ>), dim3((unsigned)nfull, (unsigned)rows), dim3(256), 0, st, d_x, (long)n, (long)nfull, bs, vec);
        VBQ_CHECK_LAUNCH("np_block_sums");
    }
    hipLaunchKernelGGL((k_np_finish<SQ>), dim3((unsigned)rows), dim3(256), 0, st, d_x, (long)n, bs, d_out);
    VBQ_CHECK_LAUNCH("np_finish");
    return VBQ_OK;
}
}  // namespace
}  // namespace vbq

extern "C" size_t vbq_numpy_sum_sq_workspace_bytes(int64_t n) {
    return n <= 0 ? 0 : (size_t)(n / vbq::kNpBlock + 1) * sizeof(float);
}

extern "C" int vbq_numpy_sum_sq_f32(const float *d_x, int64_t n, float *d_out, void *d_workspace, size_t workspace_bytes,
                                    void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n >= 0 && d_out, VBQ_ERR_INVALID_ARGUMENT, "vbq_numpy_sum_sq_f32: bad arguments");
    VBQ_REQUIRE(n == 0 || d_x, VBQ_ERR_INVALID_ARGUMENT, "vbq_numpy_sum_sq_f32: null pointer");
    VBQ_REQUIRE(n == 0 || (reinterpret_cast<uintptr_t>(d_x) & 15) == 0, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_numpy_sum_sq_f32: d_x must be 16-byte aligned");
    const size_t need = vbq_numpy_sum_sq_workspace_bytes(n);
    VBQ_REQUIRE(n == 0 || (d_workspace && workspace_bytes >= need), VBQ_ERR_WORKSPACE,
                "vbq_numpy_sum_sq_f32: workspace of %zu bytes given, %zu needed", workspace_bytes, need);
    VBQ_REQUIRE(n / kNpBlock <= 0x7fffffffLL, VBQ_ERR_UNSUPPORTED, "vbq_numpy_sum_sq_f32: more than 2^44 elements");
    return launch_np_sums<true>(d_x, n, 1, d_out, reinterpret_cast<float *>(d_workspace), reinterpret_cast<hipStream_t>(stream));
}

extern "C" size_t vbq_numpy_row_sums_workspace_bytes(int64_t n_rows, int64_t n) {
    return n_rows <= 0 || n <= 0 ? 0 : (size_t)n_rows * (size_t)(n / vbq::kNpBlock + 1) * sizeof(float);
}

extern "C" int vbq_numpy_row_sums_f32(const float *d_x, int64_t n_rows, int64_t n, float *d_out, void *d_workspace,
                                      size_t workspace_bytes, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows >= 0 && n >= 0 && n_rows <= 65535, VBQ_ERR_INVALID_ARGUMENT, "vbq_numpy_row_sums_f32: bad sizes (at most 65535 rows)");
    if (n_rows == 0) return VBQ_OK;
    VBQ_REQUIRE(d_out && (n == 0 || d_x), VBQ_ERR_INVALID_ARGUMENT, "vbq_numpy_row_sums_f32: null pointer");
    VBQ_REQUIRE((reinterpret_cast<uintptr_t>(d_x) & 3) == 0, VBQ_ERR_INVALID_ARGUMENT, "vbq_numpy_row_sums_f32: d_x must be 4-byte aligned");
    const size_t need = vbq_numpy_row_sums_workspace_bytes(n_rows, n);
    VBQ_REQUIRE(n == 0 || (d_workspace && workspace_bytes >= need), VBQ_ERR_WORKSPACE,
                "vbq_numpy_row_sums_f32: workspace of %zu bytes given, %zu needed", workspace_bytes, need);
    VBQ_REQUIRE(n / kNpBlock <= 0x7fffffffLL, VBQ_ERR_UNSUPPORTED, "vbq_numpy_row_sums_f32: more than 2^44 elements per row");
    return launch_np_sums<false>(d_x, n, n_rows, d_out, reinterpret_cast<float *>(d_workspace), reinterpret_cast<hipStream_t>(stream));
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
                              float *d_out, int32_t out_layout, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows == 0 || (d_idx && d_tab && d_out), VBQ_ERR_INVALID_ARGUMENT, "vbq_gather_f32: null pointer argument");
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_lambda >= 1 && n_lambda <= 65535 && N >= 0 && N <= 15,
                VBQ_ERR_INVALID_ARGUMENT, "vbq_gather_f32: bad sizes");
    VBQ_REQUIRE((layout == VBQ_LAYOUT_BC || layout == VBQ_LAYOUT_CB) &&
                    (out_layout == VBQ_LAYOUT_BC || out_layout == VBQ_LAYOUT_CB),
                VBQ_ERR_INVALID_ARGUMENT, "vbq_gather_f32: unknown layout %d -> %d", layout, out_layout);
    if (n_rows == 0) return VBQ_OK;
    const int64_t E = n_rows * (int64_t)n_ch;
    if (n_ch > 1 && layout != out_layout) {
        const int64_t in_rows = layout == VBQ_LAYOUT_CB ? n_ch : n_rows;
        const int64_t in_cols = layout == VBQ_LAYOUT_CB ? n_rows : n_ch;
        const int64_t tiles = ((in_cols + 31) / 32) * ((in_rows + 31) / 32);
        VBQ_REQUIRE(n_lambda <= 65535 && tiles <= 0x7fffffffll, VBQ_ERR_UNSUPPORTED,
                    "vbq_gather_f32: grid too large for the transposing form");
        hipLaunchKernelGGL(k_gather_transpose, dim3((unsigned)tiles, 1, (unsigned)n_lambda),
                           dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_idx, (long)in_rows, (long)in_cols,
                           (int)layout, (long)E, (int)n_ch, table_size(N), d_tab, (int)tab_per_lambda, d_out);
        VBQ_CHECK_LAUNCH("gather_transpose");
        return VBQ_OK;
    }
    int64_t gx = (E + 255) / 256;
    const int64_t cap = 4096 / n_lambda + 1;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(k_gather, dim3((unsigned)gx, (unsigned)n_lambda), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), d_idx, (long)n_rows, (int)n_ch, (int)layout, (long)E,
                       table_size(N), d_tab, (int)tab_per_lambda, d_out);
    VBQ_CHECK_LAUNCH("gather");
    return VBQ_OK;
}

extern "C" int vbq_rd_sums_u16(const float *d_mu, const float *d_sigma, const uint16_t *d_idx, int64_t n_rows, int32_t n_ch,
                               int32_t layout, int32_t n_lambda, int32_t N, const float *d_tab_sorted, const float *d_rate,
                               int32_t rate_per_lambda, double *d_out, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_lambda >= 1 && n_lambda <= 65535 && N >= 0 && N <= 15, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_rd_sums_u16: bad sizes");
    VBQ_REQUIRE(layout == VBQ_LAYOUT_BC || layout == VBQ_LAYOUT_CB, VBQ_ERR_INVALID_ARGUMENT, "vbq_rd_sums_u16: unknown layout %d", layout);
    if (n_rows == 0) return VBQ_OK;
    VBQ_REQUIRE(d_mu && d_sigma && d_idx && d_tab_sorted && d_out, VBQ_ERR_INVALID_ARGUMENT, "vbq_rd_sums_u16: null pointer argument");
    const int64_t E = n_rows * (int64_t)n_ch;
    const int chunks = (n_lambda + kRdChunk - 1) / kRdChunk;
    int64_t gx = (E + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 8 / chunks + 1;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(k_rd_sums, dim3((unsigned)gx, (unsigned)chunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_mu,
                       d_sigma, d_idx, (long)n_rows, (int)n_ch, (int)layout, (long)E, table_size(N), d_tab_sorted, d_rate,
                       (int)rate_per_lambda, (int)n_lambda, d_out);
    VBQ_CHECK_LAUNCH("rd_sums");
    return VBQ_OK;
}

extern "C" int vbq_transpose_planes(const void *d_in, int64_t n_batch, int64_t n_rows, int64_t n_cols, int32_t elem_bytes,
                                    void *d_out, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_batch >= 0 && n_rows >= 0 && n_cols >= 0, VBQ_ERR_INVALID_ARGUMENT, "vbq_transpose_planes: bad sizes");
    VBQ_REQUIRE(elem_bytes == 2 || elem_bytes == 4, VBQ_ERR_INVALID_ARGUMENT, "vbq_transpose_planes: elem_bytes %d (2 or 4)", elem_bytes);
    if (n_batch == 0 || n_rows == 0 || n_cols == 0) return VBQ_OK;
    VBQ_REQUIRE(d_in && d_out && d_in != d_out, VBQ_ERR_INVALID_ARGUMENT, "vbq_transpose_planes: null or aliased pointers");
    VBQ_REQUIRE(((n_rows + 63) / 64) * ((n_cols + 63) / 64) <= 0x7fffffffll && n_batch <= 65535, VBQ_ERR_UNSUPPORTED,
                "vbq_transpose_planes: more than 2^31 - 1 tiles of 64 x 64 per plane or more than 65535 planes");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (elem_bytes == 2)
        return launch_transpose_batched<uint16_t>(static_cast<const uint16_t *>(d_in), n_batch, n_rows, n_cols,
                                                  static_cast<uint16_t *>(d_out), st);
    return launch_transpose_batched<uint32_t>(static_cast<const uint32_t *>(d_in), n_batch, n_rows, n_cols,
                                              static_cast<uint32_t *>(d_out), st);
}

extern "C" int vbq_transpose_f32(const float *d_in, int64_t n_rows, int64_t n_cols, float *d_out, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows >= 0 && n_cols >= 0, VBQ_ERR_INVALID_ARGUMENT, "vbq_transpose_f32: bad sizes");
    if (n_rows == 0 || n_cols == 0) return VBQ_OK;
    VBQ_REQUIRE(d_in && d_out && d_in != d_out, VBQ_ERR_INVALID_ARGUMENT, "vbq_transpose_f32: null or aliased pointers");
    const int64_t tiles = ((n_rows + 63) / 64) * ((n_cols + 63) / 64);
    VBQ_REQUIRE(tiles <= 0x7fffffffll, VBQ_ERR_UNSUPPORTED, "vbq_transpose_f32: more than 2^31 - 1 tiles of 64 x 64");
    const bool v4 = n_rows % 4 == 0 && n_cols % 4 == 0 && ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0;
    const dim3 grid((unsigned)tiles);
    if (v4)
        hipLaunchKernelGGL(k_transpose_v4, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_in, (long)n_rows,
                           (long)n_cols, d_out);
    else
        hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_in, (long)n_rows,
                           (long)n_cols, d_out);
    VBQ_CHECK_LAUNCH("transpose");
    return VBQ_OK;
}

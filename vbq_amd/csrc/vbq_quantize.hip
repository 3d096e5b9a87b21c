// K1: the per-element rate-distortion solve of Variational Bayesian Quantization,
// all lambdas in one pass over (mu, sigma).   gfx950 / CDNA4 only.
//
// Replaces (reference file:line): quantizer.py:65-80 (interval search), :156-188
// (candidate assembly), utils.py:307-321 (distortion), utils.py:363-423 (per-lambda
// argmax), quantizer.py:135,223 (index lookup).
//
// Data layout
//   code-point table, level-major, in LDS: slot (n,i) at 2^n-1+i.  The descent
//   G_n = 2 G_{n-1} + [p_n[G_{n-1}] < z] visits one point per bit level and that point
//   is already the left or right neighbour of z on level n; the other neighbour is the
//   adjacent slot.  21 LDS reads per element (N = 10) instead of 11 searches x 10 steps.
//   Level-major (not sorted) order is what keeps those reads spread over the LDS banks:
//   in sorted order every point of levels 0..5 lives on one bank.
//   G_N is the lower bound of z in the merged sorted table, so every candidate's rank
//   index is integer arithmetic on (G_N, level, side).
// Arithmetic
//   VBQ_MODE_F32: four separately rounded f32 ops per candidate, IEEE division,
//   score = fl(s - fl(lambda*len)); candidates scanned in the reference order
//   [L_0..L_N, R_1..R_N] with a strict '>' so the first maximum wins, as np/tf argmax do.
#include <stdlib.h>

#include "vbq_common.h"

namespace vbq {
namespace {

template <int N>
struct Elem {
    float a[N + 1];   // distortion score of the left  neighbour on level n
    float b[N + 1];   // distortion score of the right neighbour on level n   (b[0] == a[0])
    uint32_t G;       // # table entries < z  (0..T)
};

// One descent through the level-major table `tb` (LDS).  Semantics of
// quantizer.py:65-80 incl. the edge padding of the search grids (:54-57): below the
// first / above the last point of a level both endpoints collapse onto it, except on the
// deepest level where the grid has no padding and the left endpoint stays one slot back.
template <int N>
__device__ __forceinline__ void search_and_score(const float *tb, float z, float sg, Elem<N> &e) {
    uint32_t g = 0;
#pragma unroll
    for (int n = 0; n <= N; ++n) {
        const int off = (1 << n) - 1;
        const int m = 1 << n;
        const uint32_t j = g;
        const float pj = tb[off + j];
        const bool below = pj < z;                       // visited point is a LEFT neighbour
        if (n == 0) {
            const float s0 = neg_half_sq_err(pj, z, sg);
            e.a[0] = s0;
            e.b[0] = s0;
        } else {
            int jo = below ? (int)j + 1 : (int)j - 1;
            jo = jo < 0 ? 0 : (jo > m - 1 ? m - 1 : jo);
            bool swap = false;
            if (n == N) {                                // no edge padding on the deepest level
                if (below && j == (uint32_t)(m - 1)) { jo = m - 2; swap = true; }
            }
            const float po = tb[off + jo];
            const float sj = neg_half_sq_err(pj, z, sg);
            const float so = neg_half_sq_err(po, z, sg);
            const bool j_is_left = below && !swap;
            e.a[n] = j_is_left ? sj : so;
            e.b[n] = j_is_left ? so : sj;
        }
        g = 2 * g + (below ? 1u : 0u);
    }
    e.G = g;
}

// Winner j in [0, 2N] (reference candidate order) -> level, slot within the level.
template <int N>
__device__ __forceinline__ void winner_slot(uint32_t G, int bj, int &lvl, uint32_t &pos) {
    const int n = bj <= N ? bj : bj - N;
    const bool right = bj > N;
    const uint32_t Gn = G >> (N - n);
    const uint32_t g = (Gn + 1) >> 1;                    // # level-n points < z
    const uint32_t m = 1u << n;
    uint32_t r = g < m - 1 ? g : m - 1;
    uint32_t l = g > 0 ? g - 1 : 0;
    if (n == N && N > 0 && g == m) l = m - 2;
    lvl = n;
    pos = right ? r : l;
}

template <int N>
__device__ __forceinline__ int argmax_scan(const Elem<N> &e, const float (&p)[N + 1]) {
    float best = __fsub_rn(e.a[0], p[0]);
    int bj = 0;
#pragma unroll
    for (int n = 1; n <= N; ++n) {
        const float sc = __fsub_rn(e.a[n], p[n]);
        const bool up = sc > best;
        best = up ? sc : best;
        bj = up ? n : bj;
    }
#pragma unroll
    for (int n = 1; n <= N; ++n) {
        const float sc = __fsub_rn(e.b[n], p[n]);
        const bool up = sc > best;
        best = up ? sc : best;
        bj = up ? N + n : bj;
    }
    return bj;
}

// f64-score variant (VBQ_MODE_F64_SCORE): score = fl64(f64(s) - fl64(lambda*len)).
template <int N>
__device__ __forceinline__ int argmax_scan_f64(const Elem<N> &e, const double (&p)[N + 1]) {
    double best = __dsub_rn((double)e.a[0], p[0]);
    int bj = 0;
#pragma unroll
    for (int n = 1; n <= N; ++n) {
        const double sc = __dsub_rn((double)e.a[n], p[n]);
        const bool up = sc > best;
        best = up ? sc : best;
        bj = up ? n : bj;
    }
#pragma unroll
    for (int n = 1; n <= N; ++n) {
        const double sc = __dsub_rn((double)e.b[n], p[n]);
        const bool up = sc > best;
        best = up ? sc : best;
        bj = up ? N + n : bj;
    }
    return bj;
}

// ------------------------------------------------------------------------------------
// penalty table: pen[l][c][n] = fl32(lambda_l) * len[l][c][n]   (len = n when raw)
// f64 variant:   pen[l][c][n] = fl64(lambda_l * len)            (raw integer lengths only)
// ------------------------------------------------------------------------------------
struct LambdaChunk {
    double lam[kMaxLambdaChunk];
};

template <typename T>
__global__ void k_prepare_penalties(LambdaChunk lc, int L, int C, int N1, const float *__restrict__ level_len,
                                    T *__restrict__ pen) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = L * C * N1;
    if (i >= total) return;
    const int n = i % N1;
    const int l = i / (N1 * C);
    if (sizeof(T) == 4) {
        const float lam = (float)lc.lam[l];
        const float len = level_len ? level_len[i] : (float)n;
        pen[i] = (T)__fmul_rn(lam, len);
    } else {
        const double len = level_len ? (double)level_len[i] : (double)n;
        pen[i] = (T)__dmul_rn(lc.lam[l], len);
    }
}

// ------------------------------------------------------------------------------------
// FLAT kernel: every element a workgroup touches uses the same table (one channel per
// blockIdx.y; C == 1 or channel-major layout).  Penalties are wave-uniform -> scalar loads.
// Each thread owns 4 consecutive elements: one 16-B load of mu and of sigma, one 8-B
// store of 4 indices per lambda.
// ------------------------------------------------------------------------------------
template <int N, typename PenT>
__global__ void __launch_bounds__(256)
k_quant_flat(const float *__restrict__ mu, const float *__restrict__ sg, long n_per_ch, long ch_stride, int C,
             const float *__restrict__ table, const PenT *__restrict__ pen, const float *__restrict__ len,
             int L, uint16_t *__restrict__ out_idx, float *__restrict__ out_zhat,
             float *__restrict__ out_bits, long E, int vec_ok) {
    constexpr int T = table_size(N);
    constexpr int N1 = N + 1;
    __shared__ float tb[T + 1];
    const int c = blockIdx.y;
    for (int i = threadIdx.x; i < T; i += blockDim.x) tb[i] = table[(long)c * T + i];
    __syncthreads();

    const long base = (long)c * ch_stride;
    const long nquads = (n_per_ch + 3) >> 2;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nquads; q += (long)gridDim.x * blockDim.x) {
        const long i0 = q * 4;
        const bool full = vec_ok && (i0 + 4 <= n_per_ch);
        float m4[4], s4[4];
        if (full) {
            const float4 mv = *reinterpret_cast<const float4 *>(mu + base + i0);
            const float4 sv = *reinterpret_cast<const float4 *>(sg + base + i0);
            m4[0] = mv.x; m4[1] = mv.y; m4[2] = mv.z; m4[3] = mv.w;
            s4[0] = sv.x; s4[1] = sv.y; s4[2] = sv.z; s4[3] = sv.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool ok = i0 + k < n_per_ch;
                m4[k] = ok ? mu[base + i0 + k] : 0.0f;
                s4[k] = ok ? sg[base + i0 + k] : 1.0f;
            }
        }
        Elem<N> el[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) search_and_score<N>(tb, m4[k], s4[k], el[k]);

        for (int l = 0; l < L; ++l) {
            const PenT *pp = pen + ((long)l * C + c) * N1;
            PenT p[N1];
#pragma unroll
            for (int n = 0; n < N1; ++n) p[n] = pp[n];
            uint32_t idx[4];
            float zh[4], bt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int bj;
                if constexpr (sizeof(PenT) == 4) bj = argmax_scan<N>(el[k], p);
                else bj = argmax_scan_f64<N>(el[k], p);
                int lvl; uint32_t pos;
                winner_slot<N>(el[k].G, bj, lvl, pos);
                idx[k] = ((2 * pos + 1) << (N - lvl)) - 1;
                if (out_zhat) zh[k] = tb[(1 << lvl) - 1 + pos];
                if (out_bits) bt[k] = len ? len[((long)l * C + c) * N1 + lvl] : (float)lvl;
            }
            const long o = (long)l * E + base + i0;
            if (full) {
                uint2 pk;
                pk.x = idx[0] | (idx[1] << 16);
                pk.y = idx[2] | (idx[3] << 16);
                *reinterpret_cast<uint2 *>(out_idx + o) = pk;
                if (out_zhat) *reinterpret_cast<float4 *>(out_zhat + o) = make_float4(zh[0], zh[1], zh[2], zh[3]);
                if (out_bits) *reinterpret_cast<float4 *>(out_bits + o) = make_float4(bt[0], bt[1], bt[2], bt[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (i0 + k < n_per_ch) {
                        out_idx[o + k] = (uint16_t)idx[k];
                        if (out_zhat) out_zhat[o + k] = zh[k];
                        if (out_bits) out_bits[o + k] = bt[k];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// TILED kernel: channel-last [rows][C] input with C > 1.  A workgroup keeps the tables of
// 16 consecutive channels in LDS (16 x 8 KB, stride T+2 words so that the same slot of
// different channels falls on different banks) plus this launch's penalty table for those
// channels, and walks down the rows.  Lane = (channel lane, row slot): a wave covers 4 rows
// x 16 channels, i.e. four 64-B input segments per load.  Each thread keeps kTiledNE rows of
// its channel in flight (independent descents hide the LDS latency); its 11 penalties per
// lambda are shared by those elements.
// ------------------------------------------------------------------------------------
constexpr int kTiledThreads = 1024;
constexpr int kTiledNE = 2;                                         // rows in flight per thread
constexpr int kRowsPerIter = (kTiledThreads / kTileChannels) * kTiledNE;

template <int N, typename PenT>
__global__ void __launch_bounds__(kTiledThreads)
k_quant_tiled(const float *__restrict__ mu, const float *__restrict__ sg, long n_rows, int C,
              const float *__restrict__ table, const PenT *__restrict__ pen, const float *__restrict__ len,
              int L, uint16_t *__restrict__ out_idx, float *__restrict__ out_zhat,
              float *__restrict__ out_bits, long E) {
    constexpr int T = table_size(N);
    constexpr int TS = T + 2;        // odd word stride: channel lane k is rotated by k banks
    constexpr int N1 = N + 1;
    constexpr int PS = (N1 + 3) & ~3;   // penalty row padded to a multiple of 4 entries
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool kPenInLds = sizeof(PenT) == 4;   // f64 penalty rows do not fit next to 16 tables
    float *tb = reinterpret_cast<float *>(smem);
    PenT *pl = reinterpret_cast<PenT *>(smem + sizeof(float) * kTileChannels * TS + 16);

    const int c0 = blockIdx.y * kTileChannels;
    const int ncg = min(kTileChannels, C - c0);
    for (int i = threadIdx.x; i < ncg * T; i += blockDim.x) {
        const int ch = i / T, s = i - ch * T;
        tb[ch * TS + s] = table[(long)(c0 + ch) * T + s];
    }
    if constexpr (kPenInLds) {
        for (int i = threadIdx.x; i < L * ncg * N1; i += blockDim.x) {
            const int n = i % N1;
            const int ch = (i / N1) % ncg;
            const int l = i / (N1 * ncg);
            pl[(l * kTileChannels + ch) * PS + n] = pen[((long)l * C + c0 + ch) * N1 + n];
        }
    }
    __syncthreads();

    const int cl = threadIdx.x & (kTileChannels - 1);
    const int slot = threadIdx.x >> 4;                     // 0..63
    if (cl >= ncg) return;
    const float *mytb = tb + cl * TS;
    const int c = c0 + cl;

    for (long rb = (long)blockIdx.x * kRowsPerIter; rb < n_rows; rb += (long)gridDim.x * kRowsPerIter) {
        Elem<N> el[kTiledNE];
        long row[kTiledNE];
#pragma unroll
        for (int k = 0; k < kTiledNE; ++k) {
            row[k] = rb + k * 64 + slot;
            const bool ok = row[k] < n_rows;
            const float m = ok ? mu[row[k] * C + c] : 0.0f;
            const float s = ok ? sg[row[k] * C + c] : 1.0f;
            search_and_score<N>(mytb, m, s, el[k]);
        }
        for (int l = 0; l < L; ++l) {
            PenT p[N1];
            const PenT *pp = kPenInLds ? pl + (l * kTileChannels + cl) * PS : pen + ((long)l * C + c) * N1;
#pragma unroll
            for (int n = 0; n < N1; ++n) p[n] = pp[n];
#pragma unroll
            for (int k = 0; k < kTiledNE; ++k) {
                int bj;
                if constexpr (sizeof(PenT) == 4) bj = argmax_scan<N>(el[k], p);
                else bj = argmax_scan_f64<N>(el[k], p);
                int lvl; uint32_t pos;
                winner_slot<N>(el[k].G, bj, lvl, pos);
                if (row[k] < n_rows) {
                    const long o = (long)l * E + row[k] * C + c;
                    out_idx[o] = (uint16_t)(((2 * pos + 1) << (N - lvl)) - 1);
                    if (out_zhat) out_zhat[o] = mytb[(1 << lvl) - 1 + pos];
                    if (out_bits) out_bits[o] = len ? len[((long)l * C + c) * N1 + lvl] : (float)lvl;
                }
            }
        }
    }
}

// ChannelwisePriorCDFQuantizer.get_all_N_bit_intervals (quantizer.py:65-80) as a result of its own: the left / right n-bit
// neighbours of every z on every level, [C][N+1][B] each -- what the reference materialises before its solve and K1 never does.
// One channel per blockIdx.y (planes), same descent as every other kernel of this file.
template <int N>
__global__ void __launch_bounds__(256)
k_intervals(const float *__restrict__ z_cb, long n_rows, const float *__restrict__ table, float *__restrict__ left,
            float *__restrict__ right) {
    constexpr int T = table_size(N);
    __shared__ float tb[T + 1];
    const int c = blockIdx.y;
    for (int i = threadIdx.x; i < T; i += blockDim.x) tb[i] = table[(long)c * T + i];
    __syncthreads();
    for (long b = (long)blockIdx.x * blockDim.x + threadIdx.x; b < n_rows; b += (long)gridDim.x * blockDim.x) {
        const float z = z_cb[(long)c * n_rows + b];
        uint32_t g = 0;
#pragma unroll
        for (int n = 0; n <= N; ++n) {
            const int off = (1 << n) - 1;
            const int m = 1 << n;
            const uint32_t j = g;
            const float pj = tb[off + j];
            const bool below = pj < z;
            float l = pj, r = pj;
            if (n > 0) {
                int jo = below ? (int)j + 1 : (int)j - 1;
                jo = jo < 0 ? 0 : (jo > m - 1 ? m - 1 : jo);
                bool swap = false;
                if (n == N && below && j == (uint32_t)(m - 1)) { jo = m - 2; swap = true; }     // no edge padding on the deepest level
                const float po = tb[off + jo];
                const bool j_is_left = below && !swap;
                l = j_is_left ? pj : po;
                r = j_is_left ? po : pj;
            }
            const long o = ((long)c * (N + 1) + n) * n_rows + b;
            left[o] = l;
            right[o] = r;
            g = 2 * g + (below ? 1u : 0u);
        }
    }
}

// VBQ_PLAIN_KERNEL=1 routes everything through the literal 21-candidate kernels (A/B checks).
inline bool force_plain_kernel() {
    static const bool v = [] { const char *e = getenv("VBQ_PLAIN_KERNEL"); return e && e[0] == '1'; }();
    return v;
}

// Rows [row_begin, row_end) of the full [n_rows x n_ch] arrays are processed; outputs keep the full arrays'
// addressing (lambda planes E = n_rows * n_ch apart).  level_counts != NULL: counting mode of the fast kernel.
template <int N, typename PenT>
int launch_quantize(const float *mu, const float *sg, int64_t n_rows, int32_t n_ch, int32_t layout,
                    const float *table, const float *level_len, const double *h_lambdas, int32_t L,
                    uint16_t *out_idx, float *out_zhat, float *out_bits, void *ws, int64_t row_begin, int64_t row_end,
                    unsigned long long *level_counts, int wg_per_cu, int reserved, hipStream_t st) {
    constexpr int N1 = N + 1;
    const int64_t E = n_rows * (int64_t)n_ch;
    const int64_t n_sub = row_end - row_begin;
    PenT *pen = reinterpret_cast<PenT *>(ws);
    const bool bc_to_cb = layout == VBQ_LAYOUT_BC_TO_CB && n_ch > 1;
    const bool flat = (n_ch == 1) || (layout == VBQ_LAYOUT_CB) || bc_to_cb;

    for (int l0 = 0; l0 < L; l0 += kMaxLambdaChunk) {
        const int Lc = (L - l0 < kMaxLambdaChunk) ? (L - l0) : kMaxLambdaChunk;
        LambdaChunk lc;
        for (int i = 0; i < kMaxLambdaChunk; ++i) lc.lam[i] = i < Lc ? h_lambdas[l0 + i] : 0.0;
        const int total = Lc * n_ch * N1;
        const float *len_c = level_len ? level_len + (int64_t)l0 * n_ch * N1 : nullptr;
        bool pen_ready = false;                             // the literal kernels read a prepared table; the fast ones do not
        auto prepare = [&]() -> int {
            if (pen_ready) return VBQ_OK;
            hipLaunchKernelGGL((k_prepare_penalties<PenT>), dim3((total + 255) / 256), dim3(256), 0, st, lc, Lc,
                               (int)n_ch, N1, len_c, pen);
            VBQ_CHECK_LAUNCH("prepare_penalties");
            pen_ready = true;
            return VBQ_OK;
        };
        Lambdas32 l32;
        for (int i = 0; i < kMaxLambdaChunk; ++i) l32.lam[i] = i < Lc ? (float)lc.lam[i] : 0.0f;

        // element offset of the first processed row: planes / one code book -> row_begin, channel-last -> row_begin * C
        const int64_t off = (flat && !bc_to_cb) ? row_begin : row_begin * n_ch;
        const int64_t off_out = flat ? row_begin : row_begin * n_ch;
        const float *mu_r = mu + off, *sg_r = sg + off;
        uint16_t *oi = out_idx ? out_idx + (int64_t)l0 * E + off_out : nullptr;
        float *oz = out_zhat ? out_zhat + (int64_t)l0 * E + off_out : nullptr;
        float *ob = out_bits ? out_bits + (int64_t)l0 * E + off_out : nullptr;
        unsigned long long *lc_out = level_counts ? level_counts + (int64_t)l0 * n_ch * N1 : nullptr;
        if (flat) {
            const int64_t n_per_ch = n_sub;                    // per channel (n_ch == 1: the whole range)
            const int64_t ch_stride = n_rows;
            // 16-byte accesses of the literal kernel (4 elements per thread) want 16-byte aligned rows and planes
            const int vec_ok = ((n_rows % 4 == 0) || n_ch == 1) &&
                               ((reinterpret_cast<uintptr_t>(mu_r) | reinterpret_cast<uintptr_t>(sg_r)) % 16 == 0) &&
                               (reinterpret_cast<uintptr_t>(oi) % 8 == 0) && (E % 4 == 0 || L == 1) &&
                               (!oz || reinterpret_cast<uintptr_t>(oz) % 16 == 0) &&
                               (!ob || reinterpret_cast<uintptr_t>(ob) % 16 == 0);
            // The fast kernels move PAIRS: 8-byte loads, 4-byte index stores, 8-byte Z_hat / length stores.  Compute queues run
            // with unaligned access enabled (ROCm's SH_MEM_CONFIG alignment mode; LLVM's amdhsa target assumes it), so element
            // alignment is all they need: odd row counts shift every other plane by half a pair and still take the paired
            // path (1.3e7 + 1 elements: K1e 52 -> 25 us per 1e6, tools/k1t_sizes.py).
            const int vec2_ok = ((reinterpret_cast<uintptr_t>(mu_r) | reinterpret_cast<uintptr_t>(sg_r)) % 4 == 0) &&
                                (reinterpret_cast<uintptr_t>(oi) % 2 == 0) && (!oz || reinterpret_cast<uintptr_t>(oz) % 4 == 0) &&
                                (!ob || reinterpret_cast<uintptr_t>(ob) % 4 == 0);
            const int64_t nquads = (n_per_ch + 3) / 4;
            int64_t gx = (nquads + 255) / 256;
            const int64_t cap = (int64_t)num_cus() * 8 / (n_ch < 8 ? n_ch : 8) + 1;   // ~8 resident workgroups per CU in total
            if (gx > cap) gx = cap;
            if (gx < 1) gx = 1;
            // The fast kernel's tie certificate needs lambda*len to be 0 or comfortably normal.
            bool fast_ok = sizeof(PenT) == 4 && !force_plain_kernel();
            // (vbq_quantize_fast.hip: the equality mask needs every lambda*len >= 2^-39 for len >= 1)
            for (int i = 0; i < Lc; ++i) fast_ok = fast_ok && (lc.lam[i] >= 1.9e-12 && lc.lam[i] <= 1.8e19);
            if constexpr (sizeof(PenT) == 4 && N == 10) {
                // first entropy-model pass (raw lengths, levels only): thresholds instead of a per-lambda loop
                if (fast_ok && lc_out && !len_c) {
                    const int r = launch_level_counts_hull10(mu_r, sg_r, n_per_ch, ch_stride, n_ch, table, lc.lam, Lc,
                                                             vec2_ok | (bc_to_cb ? 2 : 0), lc_out, reserved, st);
                    if (r == VBQ_OK) continue;
                    if (r < 0) return r;
                }
            }
            if constexpr (sizeof(PenT) == 4 && N == 10) {
                // a raw-length sweep with indices as the only output: thresholds, then a walk down the staircase
                if (fast_ok && !lc_out && !len_c && oi && !oz && !ob) {
                    const int r = launch_quant_hull_idx10(mu_r, sg_r, n_per_ch, ch_stride, n_ch, table, lc.lam, Lc,
                                                          vec2_ok | (bc_to_cb ? 2 : 0), oi, E, st);
                    if (r == VBQ_OK) continue;
                    if (r < 0) return r;
                }
            }
            if constexpr (sizeof(PenT) == 4) {
                // one to four lambdas, indices only: the descent with exact pruning (literal comparisons: any penalties)
                if (!force_plain_kernel() && !lc_out && oi && !oz && !ob && wg_per_cu == 0) {
                    const int r = launch_quant_pruned<N>(mu_r, sg_r, n_per_ch, ch_stride, n_ch, table, l32, len_c, Lc, oi, E,
                                                         vec2_ok | (bc_to_cb ? 2 : 0), st);
                    if (r == VBQ_OK) continue;
                    if (r < 0) return r;
                }
            }
            if constexpr (sizeof(PenT) == 4) {
                if (fast_ok) {
                    const int r = launch_quant_fast<N>(mu_r, sg_r, n_per_ch, ch_stride, n_ch, table, l32, len_c, Lc, oi, oz, ob,
                                                       E, vec2_ok | (bc_to_cb ? 2 : 0), lc_out, wg_per_cu, reserved, st);
                    if (r != VBQ_OK) return r;
                    continue;
                }
            }
            if (level_counts) {
                set_error("vbq_level_counts_f32 is served by the fast f32 kernel only (every lambda in [1.9e-12, 1.8e19]); "
                          "use vbq_quantize_f32 + vbq_histogram_u16 instead");
                return VBQ_ERR_UNSUPPORTED;
            }
            if (bc_to_cb) {
                set_error("vbq_quantize_f32: VBQ_LAYOUT_BC_TO_CB is served by the fast f32 kernel only (VBQ_MODE_F32, every "
                          "lambda in [1.9e-12, 1.8e19]); transpose with vbq_transpose_f32 and use VBQ_LAYOUT_CB instead");
                return VBQ_ERR_UNSUPPORTED;
            }
            if (const int r = prepare(); r != VBQ_OK) return r;
            hipLaunchKernelGGL((k_quant_flat<N, PenT>), dim3((unsigned)gx, (unsigned)n_ch), dim3(256), 0, st, mu_r, sg_r,
                               (long)n_per_ch, (long)ch_stride, (int)n_ch, table, pen, len_c, Lc, oi, oz, ob, (long)E, vec_ok);
            VBQ_CHECK_LAUNCH("quant_flat");
        } else if constexpr (N > 10) {
            set_error("vbq_quantize_f32: N=%d is built for channel-major planes only (VBQ_LAYOUT_CB, or n_ch = 1): "
                      "16 tables of %d points do not fit the LDS; transpose with vbq_transpose_f32 first", N, table_size(N));
            return VBQ_ERR_UNSUPPORTED;
        } else {
            constexpr int T = table_size(N);
            constexpr int PS = (N1 + 3) & ~3;
            const size_t lds = sizeof(float) * kTileChannels * (T + 2) + 16 +
                               (sizeof(PenT) == 4 ? sizeof(PenT) * (size_t)kMaxLambdaChunk * kTileChannels * PS : 0);
            {   // per device and cheap: set on every call (a process may drive several GPUs)
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_quant_tiled<N, PenT>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) {
                    set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
                    return VBQ_ERR_LAUNCH;
                }
            }
            const int groups = (n_ch + kTileChannels - 1) / kTileChannels;
            int64_t iters = (n_sub + kRowsPerIter - 1) / kRowsPerIter;
            int64_t gx = (256 + groups - 1) / groups;          // one workgroup per CU
            if (gx > iters) gx = iters;
            if (gx < 1) gx = 1;
            if (const int r = prepare(); r != VBQ_OK) return r;
            hipLaunchKernelGGL((k_quant_tiled<N, PenT>), dim3((unsigned)gx, (unsigned)groups), dim3(kTiledThreads),
                               lds, st, mu_r, sg_r, (long)n_sub, (int)n_ch, table, pen, len_c, Lc, oi, oz, ob, (long)E);
            VBQ_CHECK_LAUNCH("quant_tiled");
        }
    }
    return VBQ_OK;
}

}  // namespace
}  // namespace vbq

extern "C" size_t vbq_quantize_workspace_bytes(int32_t n_ch, int32_t n_lambda, int32_t N) {
    if (n_ch <= 0 || n_lambda <= 0 || N < 0) return 0;
    const int Lc = n_lambda < vbq::kMaxLambdaChunk ? n_lambda : vbq::kMaxLambdaChunk;
    return (size_t)Lc * (size_t)n_ch * (size_t)(N + 1) * sizeof(double) + 256;
}

namespace vbq {
namespace {
int quantize_entry(const char *who, const float *d_mu, const float *d_sigma, int64_t n_rows, int32_t n_ch, int32_t layout,
                   const float *d_table_lm, const float *d_level_len, const double *h_lambdas, int32_t n_lambda,
                   int32_t N, int32_t mode, uint16_t *d_out_idx, float *d_out_zhat, float *d_out_bits,
                   void *d_workspace, size_t workspace_bytes, int64_t row_begin, int64_t row_end,
                   unsigned long long *level_counts, int32_t wg_per_cu, int32_t reserved_wgs, void *stream) {
    const bool counting = level_counts != nullptr;
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_lambda >= 1, VBQ_ERR_INVALID_ARGUMENT,
                "%s: bad sizes n_rows=%lld n_ch=%d n_lambda=%d", who, (long long)n_rows, n_ch, n_lambda);
    VBQ_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= n_rows, VBQ_ERR_INVALID_ARGUMENT,
                "%s: row range [%lld, %lld) outside [0, %lld)", who, (long long)row_begin, (long long)row_end, (long long)n_rows);
    VBQ_REQUIRE(row_begin == row_end || (d_mu && d_sigma && d_table_lm && h_lambdas && (d_out_idx || counting)),
                VBQ_ERR_INVALID_ARGUMENT, "%s: null pointer argument", who);
    VBQ_REQUIRE(layout == VBQ_LAYOUT_BC || layout == VBQ_LAYOUT_CB || layout == VBQ_LAYOUT_BC_TO_CB, VBQ_ERR_INVALID_ARGUMENT,
                "%s: unknown layout %d", who, layout);
    VBQ_REQUIRE(mode == VBQ_MODE_F32 || mode == VBQ_MODE_F64_SCORE, VBQ_ERR_INVALID_ARGUMENT, "%s: unknown mode %d", who, mode);
    VBQ_REQUIRE(!(mode == VBQ_MODE_F64_SCORE && d_level_len), VBQ_ERR_UNSUPPORTED,
                "%s: VBQ_MODE_F64_SCORE supports raw integer lengths only", who);
    VBQ_REQUIRE(n_ch <= 65535, VBQ_ERR_UNSUPPORTED, "%s: n_ch=%d exceeds 65535", who, n_ch);
    VBQ_REQUIRE(wg_per_cu >= 0 && wg_per_cu <= 5, VBQ_ERR_INVALID_ARGUMENT, "%s: workgroups_per_cu=%d not in 0..5", who, wg_per_cu);
    VBQ_REQUIRE(!counting || n_ch == 1 || layout != VBQ_LAYOUT_BC, VBQ_ERR_UNSUPPORTED,
                "%s: channel-last input with n_ch > 1 is not served (use VBQ_LAYOUT_CB planes or VBQ_LAYOUT_BC_TO_CB)", who);
    const size_t need = vbq_quantize_workspace_bytes(n_ch, n_lambda, N);
    VBQ_REQUIRE(d_workspace && workspace_bytes >= need, VBQ_ERR_WORKSPACE, "%s: workspace of %zu bytes given, %zu needed", who,
                workspace_bytes, need);
    if (row_begin == row_end) return VBQ_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int reserved = reserved_wgs < 0 ? default_reserved_workgroups() : reserved_wgs;
#define VBQ_DISPATCH_N(NN)                                                                                         \
    case NN:                                                                                                       \
        return mode == VBQ_MODE_F32                                                                                \
                   ? launch_quantize<NN, float>(d_mu, d_sigma, n_rows, n_ch, layout, d_table_lm, d_level_len,      \
                                                h_lambdas, n_lambda, d_out_idx, d_out_zhat, d_out_bits,            \
                                                d_workspace, row_begin, row_end, level_counts, wg_per_cu, reserved, st) \
                   : launch_quantize<NN, double>(d_mu, d_sigma, n_rows, n_ch, layout, d_table_lm, d_level_len,     \
                                                 h_lambdas, n_lambda, d_out_idx, d_out_zhat, d_out_bits,           \
                                                 d_workspace, row_begin, row_end, level_counts, wg_per_cu, reserved, st);
    switch (N) {
        VBQ_FOR_EACH_N(VBQ_DISPATCH_N)
        default:
            set_error("%s: max_bits_per_coord N=%d not built (have 4 ... 12; only 10 with VBQ_ONLY_N10)", who, N);
            return VBQ_ERR_UNSUPPORTED;
    }
#undef VBQ_DISPATCH_N
}
}  // namespace
}  // namespace vbq

extern "C" int vbq_quantize_f32(const float *d_mu, const float *d_sigma, int64_t n_rows, int32_t n_ch,
                                int32_t layout, const float *d_table_lm, const float *d_level_len,
                                const double *h_lambdas, int32_t n_lambda, int32_t N, int32_t mode,
                                uint16_t *d_out_idx, float *d_out_zhat, float *d_out_bits, void *d_workspace,
                                size_t workspace_bytes, void *stream) {
    return vbq::quantize_entry("vbq_quantize_f32", d_mu, d_sigma, n_rows, n_ch, layout, d_table_lm, d_level_len, h_lambdas,
                               n_lambda, N, mode, d_out_idx, d_out_zhat, d_out_bits, d_workspace, workspace_bytes, 0,
                               n_rows, nullptr, 0, -1, stream);
}

extern "C" int vbq_quantize_rows_f32(const float *d_mu, const float *d_sigma, int64_t n_rows, int32_t n_ch,
                                     int32_t layout, const float *d_table_lm, const float *d_level_len,
                                     const double *h_lambdas, int32_t n_lambda, int32_t N, int32_t mode,
                                     uint16_t *d_out_idx, float *d_out_zhat, float *d_out_bits, void *d_workspace,
                                     size_t workspace_bytes, int64_t row_begin, int64_t row_end,
                                     int32_t workgroups_per_cu, int32_t reserved_workgroups, void *stream) {
    return vbq::quantize_entry("vbq_quantize_rows_f32", d_mu, d_sigma, n_rows, n_ch, layout, d_table_lm, d_level_len,
                               h_lambdas, n_lambda, N, mode, d_out_idx, d_out_zhat, d_out_bits, d_workspace, workspace_bytes,
                               row_begin, row_end, nullptr, workgroups_per_cu, reserved_workgroups, stream);
}

extern "C" int vbq_level_counts_f32(const float *d_mu, const float *d_sigma, int64_t n_rows, int32_t n_ch,
                                    int32_t layout, const float *d_table_lm, const float *d_level_len,
                                    const double *h_lambdas, int32_t n_lambda, int32_t N, int64_t *d_level_counts,
                                    void *d_workspace, size_t workspace_bytes, int32_t reserved_workgroups, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows == 0 || d_level_counts, VBQ_ERR_INVALID_ARGUMENT, "vbq_level_counts_f32: null pointer argument");
    return quantize_entry("vbq_level_counts_f32", d_mu, d_sigma, n_rows, n_ch, layout, d_table_lm, d_level_len, h_lambdas,
                          n_lambda, N, VBQ_MODE_F32, nullptr, nullptr, nullptr, d_workspace, workspace_bytes, 0, n_rows,
                          reinterpret_cast<unsigned long long *>(d_level_counts), 0, reserved_workgroups, stream);
}

extern "C" int vbq_n_bit_intervals_f32(const float *d_z_cb, int64_t n_rows, int32_t n_ch, const float *d_table_lm, int32_t N,
                                       float *d_left, float *d_right, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_ch <= 65535, VBQ_ERR_INVALID_ARGUMENT, "vbq_n_bit_intervals_f32: bad sizes");
    if (n_rows == 0) return VBQ_OK;
    VBQ_REQUIRE(d_z_cb && d_table_lm && d_left && d_right, VBQ_ERR_INVALID_ARGUMENT, "vbq_n_bit_intervals_f32: null pointer argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int64_t gx = (n_rows + 255) / 256;
    const int64_t cap = 2048 / n_ch + 1;
    if (gx > cap) gx = cap;
#define VBQ_DISPATCH_N(NN)                                                                                              \
    case NN:                                                                                                            \
        hipLaunchKernelGGL((k_intervals<NN>), dim3((unsigned)gx, (unsigned)n_ch), dim3(256), 0, st, d_z_cb, (long)n_rows, \
                           d_table_lm, d_left, d_right);                                                                \
        break;
    switch (N) {
        VBQ_FOR_EACH_N(VBQ_DISPATCH_N)
        default:
            set_error("vbq_n_bit_intervals_f32: max_bits_per_coord N=%d not built", N);
            return VBQ_ERR_UNSUPPORTED;
    }
#undef VBQ_DISPATCH_N
    VBQ_CHECK_LAUNCH("n_bit_intervals");
    return VBQ_OK;
}

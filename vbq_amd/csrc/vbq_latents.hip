// The per-image call of the reference's evaluation loop -- ChannelwisePriorCDFQuantizer.compress_latents,
// img-compression/quantizer.py:190-240, called once per image by utils.py:542-554 -- as ONE C call and three launches:
//
//   k_prep_planes       channel-last means / spreads [B][C] -> channel-major planes [C][B] (quantizer.py:163-164,223),
//                       with sigma = sqrt(exp(logvar)) folded in when the caller hands the log-variances as they come out of
//                       the encoder (quantizer.py:197,202: `tf.exp(posterior_logvars) ** 0.5`): sqrtf is the IEEE root
//                       (__fsqrt_rn is NOT on gfx950) and expf is the device library's exp -- the numbers torch's
//                       `exp(x) ** 0.5` gives on this device for every one of the 2^32 float32 inputs
//                       (tests/test_gpu_api.py::test_logvar_to_sigma_is_torchs_for_every_float32)
//   K1 / K1e / K1p      the solve on planes (vbq_quantize_f32)
//   k_gather_latents    ONE pass over the rank indices [L][C][B]: Z_hat = sorted table[c][q]  (quantizer.py:224-225),
//                       raw_num_bits = level(q) or level_len[l][c][level(q)] (:171-175,186-188), num_bits =
//                       entropy_model[l][c][q] (:226-228), all written channel-last [L][B][C] (the np.reshape of :237),
//                       optionally the indices themselves channel-last.  From two images up the Z_hat and num_bits lookups
//                       go to k_lookup_lds (16 channel tables resident in the LDS) and this pass keeps the rest.
// and, at the end of the file, the whole two-pass build of quantizer.py:82-150 as one C call (vbq_build_entropy_models_f32).
//
// The level of rank index q is N - ctz(q + 1) (slot (n, i) <-> rank (2i + 1) 2^(N-n) - 1), so no per-rank length table is
// built or read.
#include <stdlib.h>

#include <atomic>

#include <type_traits>

#include "vbq_common.h"

namespace vbq {
namespace {

// ---------------------------------------------------------------------------- planes from channel-last latents
// in [rows][cols] -> out [cols][rows], 64 x 64 tiles through LDS; blockIdx.y = 0: means, 1: spreads (sqrt when asked).
__global__ void __launch_bounds__(256)
k_prep_planes(const float *__restrict__ in0, const float *__restrict__ in1, long rows, long cols, float *__restrict__ out0,
              float *__restrict__ out1, int sqrt1, int v4) {
    __shared__ float tile[64][65];
    const bool second = blockIdx.y == 1;
    const float *in = second ? in1 : in0;
    float *out = second ? out1 : out0;
    const int kind = second ? sqrt1 : 0;                         // 0: copy, 1: sqrt(x), 2: sqrt(exp(x))
    auto f = [kind](float x) { return kind == 0 ? x : sqrtf(kind == 2 ? expf(x) : x); };
    const long ctiles = (cols + 63) / 64;
    const long r0 = ((long)blockIdx.x / ctiles) * 64, c0 = ((long)blockIdx.x % ctiles) * 64;
    if (v4 && r0 + 64 <= rows && c0 + 64 <= cols) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k, lr = i >> 4, lc = (i & 15) * 4;
            v[k] = *reinterpret_cast<const float4 *>(in + (r0 + lr) * cols + c0 + lc);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k, lr = i >> 4, lc = (i & 15) * 4;
            if (kind) { v[k].x = f(v[k].x); v[k].y = f(v[k].y); v[k].z = f(v[k].z); v[k].w = f(v[k].w); }
            tile[lr][lc] = v[k].x; tile[lr][lc + 1] = v[k].y; tile[lr][lc + 2] = v[k].z; tile[lr][lc + 3] = v[k].w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k, lc = i >> 4, lr = (i & 15) * 4;
            *reinterpret_cast<float4 *>(out + (c0 + lc) * rows + r0 + lr) =
                make_float4(tile[lr][lc], tile[lr + 1][lc], tile[lr + 2][lc], tile[lr + 3][lc]);
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const long r = r0 + ty + 4 * k, c = c0 + tx;
        if (r < rows && c < cols) {
            const float x = in[r * cols + c];
            tile[ty + 4 * k][tx] = f(x);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const long c = c0 + ty + 4 * k, r = r0 + tx;
        if (r < rows && c < cols) out[c * rows + r] = tile[tx][ty + 4 * k];
    }
}

// ---------------------------------------------------------------------------- fused lookups of one solve
// idx planes [L][C][B] -> up to four channel-last outputs [L][B][C].  Tile = 64 channels x 32 rows: the index reads are
// 64-byte runs along a plane, every output row segment is 256 bytes (128 for the u16 indices).
constexpr int kGlRows = 32, kGlCh = 64;
template <int N>
__global__ void __launch_bounds__(256)
k_gather_latents(const uint16_t *__restrict__ idx, long B, int C, const float *__restrict__ tab_sorted,
                 const float *__restrict__ level_len, const float *__restrict__ models, float *__restrict__ out_z,
                 float *__restrict__ out_raw, int raw_as_int, float *__restrict__ out_nb, uint16_t *__restrict__ out_idx) {
    constexpr int T = table_size(N), N1 = N + 1;
    __shared__ float tz[kGlCh][kGlRows + 1], tr[kGlCh][kGlRows + 1], tn[kGlCh][kGlRows + 1];
    __shared__ uint16_t tq[kGlCh][kGlRows + 2];
    const int l = blockIdx.y;
    const long ctiles = (C + kGlCh - 1) / kGlCh;
    const long r0 = ((long)blockIdx.x / ctiles) * kGlRows;
    const int c0 = (int)((long)blockIdx.x % ctiles) * kGlCh;
    const long E = B * (long)C;
    {
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 rows x 8 channels per step
        const long r = r0 + tx;
#pragma unroll
        for (int k = 0; k < kGlCh / 8; ++k) {
            const int lc = ty + 8 * k, c = c0 + lc;
            if (r < B && c < C) {
                const int q = min((int)idx[((long)l * C + c) * B + r], T - 1);      // foreign indices >= T stay inside the tables
                const int lvl = N - __builtin_ctz((unsigned)q + 1u);
                if (out_z) tz[lc][tx] = tab_sorted[(long)c * T + q];
                if (out_raw) tr[lc][tx] = level_len ? level_len[((long)l * C + c) * N1 + lvl]
                                                    : (raw_as_int ? __int_as_float(lvl) : (float)lvl);
                if (out_nb) tn[lc][tx] = models[((long)l * C + c) * T + q];
                if (out_idx) tq[lc][tx] = (uint16_t)q;
            }
        }
    }
    __syncthreads();
    {
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;     // 64 channels x 4 rows per step
        const int c = c0 + tx;
#pragma unroll
        for (int k = 0; k < kGlRows / 4; ++k) {
            const int lr = ty + 4 * k;
            const long r = r0 + lr;
            if (r < B && c < C) {
                const long o = (long)l * E + r * C + c;
                if (out_z) out_z[o] = tz[tx][lr];
                if (out_raw) out_raw[o] = tr[tx][lr];
                if (out_nb) out_nb[o] = tn[tx][lr];
                if (out_idx) out_idx[o] = tq[tx][lr];
            }
        }
    }
}

// The same with 16-byte accesses on both sides (B a multiple of 8, C a multiple of 4, 16-byte aligned pointers): tile = 64
// channels x 64 rows; a thread reads two runs of 8 consecutive indices of one channel, issues all its table lookups at once
// (up to 48 independent loads in flight: the lookups are latency-bound, the tables sit in L2), and the outputs go through ONE
// LDS tile, one after the other, leaving as 16-byte stores of 4 consecutive channels.  The one-tile form keeps the LDS at
// 17 KB per workgroup so that enough waves are resident to cover the lookups (a tile per output: 58 KB, 2 workgroups per CU).
template <int N>
__global__ void __launch_bounds__(256)
k_gather_latents_vec(const uint16_t *__restrict__ idx, long B, int C, const float *__restrict__ tab_sorted,
                     const float *__restrict__ level_len, const float *__restrict__ models, float *__restrict__ out_z,
                     float *__restrict__ out_raw, int raw_as_int, float *__restrict__ out_nb, uint16_t *__restrict__ out_idx) {
    constexpr int T = table_size(N), N1 = N + 1;
    __shared__ float tile[64][65];
    const int l = blockIdx.y;
    const long ctiles = (C + 63) / 64;
    const long r0 = ((long)blockIdx.x / ctiles) * 64;
    const int c0 = (int)((long)blockIdx.x % ctiles) * 64;
    const long E = B * (long)C;
    int q[2][8];
    bool live[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int v = threadIdx.x + 256 * k, lc = v >> 3, rg = (v & 7) * 8;
        const int c = c0 + lc;
        const long r = r0 + rg;
        live[k] = c < C && r < B;                               // B % 8 == 0: a run of 8 rows is inside or outside as a whole
        uint4 w = make_uint4(0, 0, 0, 0);
        if (live[k]) w = *reinterpret_cast<const uint4 *>(idx + ((long)l * C + c) * B + r);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) q[k][j] = min((int)((ww[j >> 1] >> (16 * (j & 1))) & 0xffffu), T - 1);
    }
    float vz[2][8], vr[2][8], vn[2][8];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int v = threadIdx.x + 256 * k, lc = v >> 3;
        const int c = min(c0 + lc, C - 1);
        const float *ts = tab_sorted + (long)c * T;
        const float *ll = level_len + ((long)l * C + c) * N1;
        const float *mm = models + ((long)l * C + c) * T;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lvl = N - __builtin_ctz((unsigned)q[k][j] + 1u);
            if (out_z) vz[k][j] = ts[q[k][j]];
            if (out_raw) vr[k][j] = level_len ? ll[lvl] : (raw_as_int ? __int_as_float(lvl) : (float)lvl);
            if (out_nb) vn[k][j] = mm[q[k][j]];
        }
    }
    auto emit = [&](const float (&val)[2][8], float *__restrict__ out) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int v = threadIdx.x + 256 * k, lc = v >> 3, rg = (v & 7) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) tile[lc][rg + j] = val[k][j];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = threadIdx.x + 256 * k, lr = v >> 4, lc = (v & 15) * 4;       // 16 float4 per output row
            const long r = r0 + lr;
            const int c = c0 + lc;
            if (r < B && c < C)                                 // C % 4 == 0: 4 channels are inside or outside as a whole
                *reinterpret_cast<float4 *>(out + (long)l * E + r * C + c) =
                    make_float4(tile[lc][lr], tile[lc + 1][lr], tile[lc + 2][lr], tile[lc + 3][lr]);
        }
        __syncthreads();
    };
    if (out_z) emit(vz, out_z);
    if (out_raw) emit(vr, out_raw);
    if (out_nb) emit(vn, out_nb);
    if (out_idx) {
        uint16_t *t16 = reinterpret_cast<uint16_t *>(&tile[0][0]);                     // [64][66] u16 inside the same block
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int v = threadIdx.x + 256 * k, lc = v >> 3, rg = (v & 7) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) t16[lc * 66 + rg + j] = (uint16_t)q[k][j];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = threadIdx.x + 256 * k, lr = v >> 4, lc = (v & 15) * 4;
            const long r = r0 + lr;
            const int c = c0 + lc;
            if (r < B && c < C) {
                uint2 w;
                w.x = (uint32_t)t16[lc * 66 + lr] | ((uint32_t)t16[(lc + 1) * 66 + lr] << 16);
                w.y = (uint32_t)t16[(lc + 2) * 66 + lr] | ((uint32_t)t16[(lc + 3) * 66 + lr] << 16);
                *reinterpret_cast<uint2 *>(out_idx + (long)l * E + r * C + c) = w;
            }
        }
    }
    (void)live;
}

// ---------------------------------------------------------------------------- lookups out of LDS-resident tables
// k_gather_latents reads its tables through L2, and every lane of a lookup hits its own 64-byte sector: 3.2e11 lookups/s
// chip-wide whatever the tile shape (EXPERIMENTS.md).  For LARGE batches the tables can live in the LDS instead: one workgroup
// of 1024 threads owns 16 channels -- 16 tables of 8 KB = 128 KB of the CU's 160 KB -- wave w serves channel c0 + w, a lane
// reads four consecutive indices of its channel (8 bytes), looks them up in the LDS and parks the four values as one 16-byte
// LDS write in a [16 channels][256 rows] tile; after a barrier every thread takes four channels of one row out of the tile and
// stores 16 bytes: the output rows leave as 64-byte segments (16 channels x f32).  Two kinds of pass:
//   per_lambda == 0   tab = d_table_sorted [C][T]: the tables do not depend on lambda -- a workgroup keeps them for ALL lambdas
//                     of its row range (blockIdx.y splits the rows); outputs Z_hat, optionally raw_num_bits
//   per_lambda == 1   tab = d_models [L][C][T]: a workgroup owns ONE lambda (blockIdx.y) and all rows; output num_bits
// Worth it when a staged table entry is looked up many times (the launcher's rule); N <= 10, B % 4 == 0, C % 4 == 0.
// Two tiles of [16][256] floats, used in turn (one barrier per emitted output instead of two); no padding fits beside the tables,
// so the rows of channels 4..7 and 12..15 are stored with bit 4 flipped: the row-wise reads of the four channel groups then fall
// on banks 0-15 / 16-31 / 0-15 / 16-31 (2-way, the minimum for 64 lanes) and the 16-byte writes stay whole.
#ifndef VBQ_LDS_AHEAD
#define VBQ_LDS_AHEAD 2
#endif
constexpr int kLdsCh = 16, kLdsRows = 256, kLdsPitch = kLdsRows, kLdsAhead = VBQ_LDS_AHEAD;
// From where the LDS form is taken (C = 256, tools/gather_bench.py and the bench's per-image call).  With uniform random indices it
// wins from one image on (1 536 rows x 16 lambdas: num_bits 22.8 against 32.8 us); with the indices of a real solve -- a few hot
// code points per lambda, whose sectors the L2 form's lanes share -- one image is faster in ONE launch of the L2 form (47.7
// against 51.0 us for the three outputs), two images and more in the LDS form (74.6 against 107 us; Kodak-24 x 32 Z_hat, real
// indices, inside the facade: 1.23 -> 0.84 ms).
constexpr int64_t kLdsMinLookupsZ = 9 << 14;         // lambdas x rows per channel table (Z_hat, with raw_num_bits riding along): from about six Kodak
                                                     // images x 16 lambdas up (round 6, tools/image_lookup_modes.py: 2 / 4 images are 7-8 % faster with the table in L2)
constexpr int64_t kLdsMinLookupsNb = 3 << 10;        // rows per (lambda, channel) table (num_bits)
template <int N>
__global__ void __launch_bounds__(1024)
k_lookup_lds(const uint16_t *__restrict__ idx, long B, int C, int L, const float *__restrict__ tab, int per_lambda,
             float *__restrict__ out_a, long rows_per_split) {
    constexpr int T = table_size(N), TP = T;
    extern __shared__ float lds_f[];
    float *tabs = lds_f;                                       // [16][T]
    float *tiles = lds_f + ((kLdsCh * TP + 3) & ~3);           // two tiles [16][256], 16-byte aligned
    int flip = 0;
    // Workgroups go to the 8 XCDs round-robin by their linear number.  A workgroup writes 64-byte halves of 128-byte lines; with
    // the channel groups renumbered inside every block of 16 so that groups 2k and 2k + 1 get numbers 8 apart, the two halves of a
    // line are written by workgroups of ONE XCD, walking the same rows in the same order, and meet in that XCD's L2.
    int grp = blockIdx.x;
    if ((gridDim.x & 15) == 0) grp = (grp & ~15) + ((grp & 7) << 1) + ((grp >> 3) & 1);
    const int c0 = grp * kLdsCh;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = c0 + w;
    const bool ch_ok = c < C;
    const int l_begin = per_lambda ? (int)blockIdx.y : 0, l_end = per_lambda ? (int)blockIdx.y + 1 : L;
    const long r_begin = per_lambda ? 0 : (long)blockIdx.y * rows_per_split;
    const long r_end = per_lambda ? B : (r_begin + rows_per_split < B ? r_begin + rows_per_split : B);
    const long E = B * (long)C;
    for (int l = l_begin; l < l_end; ++l) {
        if (per_lambda || l == l_begin) {                       // stage the 16 tables (coalesced rows of 8 KB)
            if (per_lambda && l != l_begin) __syncthreads();    // (a workgroup owns one lambda in that mode: never taken today)
            for (int i = threadIdx.x; i < kLdsCh * TP; i += blockDim.x) {
                const int tc = i / TP, e = i - tc * TP;
                const int cc = c0 + tc < C ? c0 + tc : C - 1;
                tabs[i] = tab[((per_lambda ? (long)l * C : 0) + cc) * T + e];
            }
            __syncthreads();
        }
        const uint16_t *src = idx + ((long)l * C + (ch_ok ? c : 0)) * B;
        // B % 4 == 0: four indices are inside or outside together.  The load is UNCONDITIONAL (rows beyond the range re-read the
        // channel's last four indices; their results are never stored): a branch around it makes the compiler wait for every
        // outstanding store before every block (s_waitcnt vmcnt(0) at the loop head) -- stores and lookups then take turns.
        auto load4 = [&](long r0) {
            long r = r0 + 4 * lane;
            r = r < B - 4 ? r : B - 4;
            return *reinterpret_cast<const uint2 *>(src + r);
        };
        // One block of 256 rows: four lookups per lane, the values crossed to row order through an LDS tile, 16-byte stores.
        // FULL: every row and every channel of the block exists -- the stores are unconditional (no branch in the block).
        auto block = [&](long r0, uint2 q2, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            const int q[4] = {(int)(q2.x & 0xffffu), (int)(q2.x >> 16), (int)(q2.y & 0xffffu), (int)(q2.y >> 16)};
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = tabs[w * TP + min(q[j], T - 1)];
            float *tile = tiles + flip * (kLdsCh * kLdsPitch);
            flip ^= 1;                                          // the tile written two blocks ago is free again: a barrier lies between
            *reinterpret_cast<float4 *>(tile + w * kLdsPitch + ((4 * lane) ^ ((w & 4) << 2))) = make_float4(v[0], v[1], v[2], v[3]);
            __syncthreads();
            const int row = threadIdx.x >> 2, g = threadIdx.x & 3, ch4 = g * 4;
            const int rs = row ^ ((g & 1) << 4);                // (ch4 & 4) << 2: the flip of this channel group's rows
            const long rr = r0 + row;
            if (FULL || (rr < r_end && c0 + ch4 < C))            // C % 4 == 0: four channels are inside or outside together
                *reinterpret_cast<float4 *>(out_a + (long)l * E + rr * C + c0 + ch4) =
                    make_float4(tile[ch4 * kLdsPitch + rs], tile[(ch4 + 1) * kLdsPitch + rs], tile[(ch4 + 2) * kLdsPitch + rs],
                                tile[(ch4 + 3) * kLdsPitch + rs]);
        };
        long r0 = r_begin;
        if (c0 + kLdsCh <= C) {
            // Whole blocks, kLdsAhead at a time, with that many blocks of indices in flight per wave.  Every slot of the ring has
            // its own registers and is refilled right after it is read, and nothing in the body is conditional: with a branch
            // around the loads or the stores the compiler emits s_waitcnt vmcnt(0) at the head of EVERY block, and a block's
            // lookups wait for the stores of the block before (measured with the stores switched off: 346 of 552 us were the
            // rest, the two did not overlap at all).  Like this the drain comes once per kLdsAhead blocks: Kodak-24 x 32 Z_hat
            // 510 -> 456 us with two blocks (four: 548, eight: 584 -- longer bodies, more registers).
            uint2 qa[kLdsAhead];
#pragma unroll
            for (int i = 0; i < kLdsAhead; ++i) qa[i] = load4(r0 + (long)i * kLdsRows);
            for (; r0 + (long)kLdsAhead * kLdsRows <= r_end; r0 += (long)kLdsAhead * kLdsRows) {
#pragma unroll
                for (int i = 0; i < kLdsAhead; ++i) {
                    const uint2 q2 = qa[i];
                    qa[i] = load4(r0 + (long)(kLdsAhead + i) * kLdsRows);
                    block(r0 + (long)i * kLdsRows, q2, std::true_type{});
                }
            }
        }
        for (; r0 < r_end; r0 += kLdsRows) block(r0, load4(r0), std::false_type{});       // the ragged rest
    }
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// LDS passes (k_lookup_lds) for the outputs they pay for; returns through *did_z / *did_nb which outputs they produced.
template <int N>
int lookup_lds_passes(const uint16_t *idx, int64_t B, int32_t C, int32_t L, const float *tab_sorted, const float *models,
                      float *out_z, float *out_nb, bool *did_z, bool *did_nb, hipStream_t st) {
    *did_z = *did_nb = false;
    if constexpr (N > 10) {
        return VBQ_OK;
    } else {
        static const int mode = [] { const char *e = getenv("VBQ_LOOKUP_LDS"); return e ? atoi(e) : -1; }();   // A/B: 0 never, 1 always
        const uintptr_t al = reinterpret_cast<uintptr_t>(out_z) | reinterpret_cast<uintptr_t>(out_nb);
        if (mode == 0 || B % 4 != 0 || C % 4 != 0 || (al & 15) != 0 || (reinterpret_cast<uintptr_t>(idx) & 7) != 0) return VBQ_OK;
        constexpr int TP = table_size(N);
        const size_t lds = sizeof(float) * (size_t)(((kLdsCh * TP + 3) & ~3) + 2 * kLdsCh * kLdsPitch);
        const int groups = (C + kLdsCh - 1) / kLdsCh;
        // a staged table entry must be looked up often enough to pay for its staging: measured break-even (tools/gather_bench.py)
        const bool want_z = out_z != nullptr && (mode == 1 || mode == 3 || (mode < 0 && (int64_t)L * B >= kLdsMinLookupsZ));
        const bool want_nb = out_nb != nullptr && (mode == 1 || mode == 2 || (mode < 0 && B >= kLdsMinLookupsNb));
        if (!want_z && !want_nb) return VBQ_OK;
        // The LDS opt-in of the kernel, once per (device, N): 1 = granted, 2 = refused (a device or partition mode with less LDS
        // to opt into).  The passes are an optimisation: refused, or with a grid beyond the launch limits, the outputs stay with
        // the generic kernel, which can always produce them.
        static std::atomic<signed char> lds_state[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
            (void)hipGetLastError();
            return VBQ_OK;
        }
        signed char state = lds_state[dev].load(std::memory_order_relaxed);
        if (state == 0) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lookup_lds<N>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) (void)hipGetLastError();
            state = e == hipSuccess ? 1 : 2;
            lds_state[dev].store(state, std::memory_order_relaxed);
        }
        if (state != 1) return VBQ_OK;
        if (want_z) {
            int64_t splits = (2 * (int64_t)num_cus() + groups - 1) / groups;                 // about two workgroups per CU in all
            const int64_t blocks = (B + kLdsRows - 1) / kLdsRows;
            if (splits > blocks) splits = blocks;
            if (splits < 1) splits = 1;
            const int64_t per = ((blocks + splits - 1) / splits) * kLdsRows;
            splits = (B + per - 1) / per;
            if (splits > 65535) return VBQ_OK;                 // (not reachable with ~2 workgroups per CU; the generic kernel serves it)
            hipLaunchKernelGGL((k_lookup_lds<N>), dim3((unsigned)groups, (unsigned)splits), dim3(1024), lds, st, idx, (long)B, (int)C, (int)L,
                               tab_sorted, 0, out_z, (long)per);
            VBQ_CHECK_LAUNCH("lookup_lds (sorted table)");
            *did_z = true;
        }
        if (want_nb && L <= 65535) {
            hipLaunchKernelGGL((k_lookup_lds<N>), dim3((unsigned)groups, (unsigned)L), dim3(1024), lds, st, idx, (long)B, (int)C, (int)L,
                               models, 1, out_nb, (long)B);
            VBQ_CHECK_LAUNCH("lookup_lds (entropy models)");
            *did_nb = true;
        }
        return VBQ_OK;
    }
}

int gather_latents(const uint16_t *idx, int64_t B, int32_t C, int32_t L, int32_t N, const float *tab_sorted, const float *level_len,
                   const float *models, float *out_z, void *out_raw, float *out_nb, uint16_t *out_idx, hipStream_t st) {
    {   // large batches: tables in the LDS for the outputs that pay for it, the generic kernel for what is left
        bool did_z = false, did_nb = false;
        int rc = VBQ_OK;
#define VBQ_DISPATCH_N(NN)                                                                                              \
    case NN:                                                                                                            \
        rc = lookup_lds_passes<NN>(idx, B, C, L, tab_sorted, models, out_z, out_nb, &did_z, &did_nb, st);             \
        break;
        switch (N) {
            VBQ_FOR_EACH_N(VBQ_DISPATCH_N)
            default: break;
        }
#undef VBQ_DISPATCH_N
        if (rc != VBQ_OK) return rc;
        if (did_z) out_z = nullptr;                            // (raw_num_bits needs no table worth the LDS: the generic pass writes it at its 5 TB/s)
        if (did_nb) out_nb = nullptr;
        if (!out_z && !out_raw && !out_nb && !out_idx) return VBQ_OK;
    }
    const uintptr_t all = reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(out_z) | reinterpret_cast<uintptr_t>(out_raw) |
                          reinterpret_cast<uintptr_t>(out_nb) | reinterpret_cast<uintptr_t>(out_idx);
    static const bool scalar_only = [] { const char *e = getenv("VBQ_GATHER_SCALAR"); return e && e[0] == '1'; }();   // A/B timing
    const bool vec = !scalar_only && B % 8 == 0 && C % 4 == 0 && (all & 15) == 0;
    const int64_t tiles = vec ? ((B + 63) / 64) * ((C + 63) / 64) : ((B + kGlRows - 1) / kGlRows) * ((C + kGlCh - 1) / kGlCh);
    VBQ_REQUIRE(tiles <= 0x7fffffffll && L <= 65535, VBQ_ERR_UNSUPPORTED, "vbq_gather_latents_u16: grid too large");
    const dim3 grid((unsigned)tiles, (unsigned)L);
    const int raw_as_int = level_len == nullptr;
#define VBQ_DISPATCH_N(NN)                                                                                                  \
    case NN:                                                                                                                \
        if (vec)                                                                                                            \
            hipLaunchKernelGGL((k_gather_latents_vec<NN>), grid, dim3(256), 0, st, idx, (long)B, (int)C, tab_sorted, level_len, \
                               models, out_z, static_cast<float *>(out_raw), raw_as_int, out_nb, out_idx);                  \
        else                                                                                                                \
            hipLaunchKernelGGL((k_gather_latents<NN>), grid, dim3(256), 0, st, idx, (long)B, (int)C, tab_sorted, level_len, \
                               models, out_z, static_cast<float *>(out_raw), raw_as_int, out_nb, out_idx);                  \
        break;
    switch (N) {
        VBQ_FOR_EACH_N(VBQ_DISPATCH_N)
        default:
            set_error("vbq_gather_latents_u16: max_bits_per_coord N=%d not built", N);
            return VBQ_ERR_UNSUPPORTED;
    }
#undef VBQ_DISPATCH_N
    VBQ_CHECK_LAUNCH("gather_latents");
    return VBQ_OK;
}

}  // namespace
}  // namespace vbq

extern "C" int vbq_gather_latents_u16(const uint16_t *d_idx_planes, int64_t n_rows, int32_t n_ch, int32_t n_lambda, int32_t N,
                                      const float *d_table_sorted, const float *d_level_len, const float *d_models,
                                      float *d_out_zhat, void *d_out_raw_bits, float *d_out_num_bits, uint16_t *d_out_idx,
                                      void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_lambda >= 1 && N >= 0 && N <= 15, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_gather_latents_u16: bad sizes");
    if (n_rows == 0) return VBQ_OK;
    VBQ_REQUIRE(d_idx_planes && (d_out_zhat || d_out_raw_bits || d_out_num_bits || d_out_idx), VBQ_ERR_INVALID_ARGUMENT,
                "vbq_gather_latents_u16: null pointer argument");
    VBQ_REQUIRE((!d_out_zhat || d_table_sorted) && (!d_out_num_bits || d_models), VBQ_ERR_INVALID_ARGUMENT,
                "vbq_gather_latents_u16: an output without its table");
    return gather_latents(d_idx_planes, n_rows, n_ch, n_lambda, N, d_table_sorted, d_level_len, d_models, d_out_zhat, d_out_raw_bits,
                          d_out_num_bits, d_out_idx, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int vbq_prep_planes_f32(const float *d_means_bc, const float *d_spread_bc, int32_t spread_kind, int64_t n_rows,
                                   int32_t n_ch, float *d_mu_cb, float *d_sigma_cb, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1, VBQ_ERR_INVALID_ARGUMENT, "vbq_prep_planes_f32: bad sizes");
    VBQ_REQUIRE(spread_kind >= VBQ_SPREAD_SIGMA && spread_kind <= VBQ_SPREAD_LOGVAR, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_prep_planes_f32: unknown spread_kind %d", spread_kind);
    if (n_rows == 0) return VBQ_OK;
    VBQ_REQUIRE(d_means_bc && d_spread_bc && d_mu_cb && d_sigma_cb && d_means_bc != d_mu_cb && d_spread_bc != d_sigma_cb,
                VBQ_ERR_INVALID_ARGUMENT, "vbq_prep_planes_f32: null or aliased pointers");
    const int64_t tiles = ((n_rows + 63) / 64) * (((int64_t)n_ch + 63) / 64);
    VBQ_REQUIRE(tiles <= 0x7fffffffll, VBQ_ERR_UNSUPPORTED, "vbq_prep_planes_f32: more than 2^31 - 1 tiles of 64 x 64");
    const uintptr_t all = reinterpret_cast<uintptr_t>(d_means_bc) | reinterpret_cast<uintptr_t>(d_spread_bc) |
                          reinterpret_cast<uintptr_t>(d_mu_cb) | reinterpret_cast<uintptr_t>(d_sigma_cb);
    const int v4 = n_rows % 4 == 0 && n_ch % 4 == 0 && (all & 15) == 0;
    hipLaunchKernelGGL(k_prep_planes, dim3((unsigned)tiles, 2), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_means_bc,
                       d_spread_bc, (long)n_rows, (long)n_ch, d_mu_cb, d_sigma_cb, (int)spread_kind, v4);
    VBQ_CHECK_LAUNCH("prep_planes");
    return VBQ_OK;
}

extern "C" size_t vbq_compress_latents_workspace_bytes(int64_t n_rows, int32_t n_ch, int32_t n_lambda, int32_t N) {
    if (n_rows < 0 || n_ch <= 0 || n_lambda <= 0 || N < 0) return 0;
    const size_t E = (size_t)n_rows * (size_t)n_ch;
    return 2 * vbq::align256(E * sizeof(float)) + vbq::align256((size_t)n_lambda * E * sizeof(uint16_t)) +
           vbq::align256(vbq_quantize_workspace_bytes(n_ch, n_lambda, N));
}

extern "C" int vbq_compress_latents_f32(const float *d_means_bc, const float *d_spread_bc, int32_t spread_kind,
                                        int64_t n_rows, int32_t n_ch, const float *d_table_lm, const float *d_table_sorted,
                                        const float *d_level_len, const float *d_models, const double *h_lambdas,
                                        int32_t n_lambda, int32_t N, float *d_out_zhat, void *d_out_raw_bits,
                                        float *d_out_num_bits, void *d_workspace, size_t workspace_bytes, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows >= 0 && n_ch >= 1 && n_lambda >= 1, VBQ_ERR_INVALID_ARGUMENT, "vbq_compress_latents_f32: bad sizes");
    if (n_rows == 0) return VBQ_OK;
    const size_t need = vbq_compress_latents_workspace_bytes(n_rows, n_ch, n_lambda, N);
    VBQ_REQUIRE(d_workspace && workspace_bytes >= need, VBQ_ERR_WORKSPACE, "vbq_compress_latents_f32: workspace of %zu bytes given, %zu needed",
                workspace_bytes, need);
    VBQ_REQUIRE((reinterpret_cast<uintptr_t>(d_workspace) & 255) == 0, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_compress_latents_f32: workspace must be 256-byte aligned");
    const size_t E = (size_t)n_rows * (size_t)n_ch;
    char *w = static_cast<char *>(d_workspace);
    float *mu_cb = reinterpret_cast<float *>(w);
    w += align256(E * sizeof(float));
    float *sg_cb = reinterpret_cast<float *>(w);
    w += align256(E * sizeof(float));
    uint16_t *idx = reinterpret_cast<uint16_t *>(w);
    w += align256((size_t)n_lambda * E * sizeof(uint16_t));
    const size_t qws = vbq_quantize_workspace_bytes(n_ch, n_lambda, N);
    int rc = vbq_prep_planes_f32(d_means_bc, d_spread_bc, spread_kind, n_rows, n_ch, mu_cb, sg_cb, stream);
    if (rc != VBQ_OK) return rc;
    rc = vbq_quantize_f32(mu_cb, sg_cb, n_rows, n_ch, VBQ_LAYOUT_CB, d_table_lm, d_level_len, h_lambdas, n_lambda, N, VBQ_MODE_F32,
                          idx, nullptr, nullptr, w, qws, stream);
    if (rc != VBQ_OK) return rc;
    return vbq_gather_latents_u16(idx, n_rows, n_ch, n_lambda, N, d_table_sorted, d_level_len, d_models, d_out_zhat, d_out_raw_bits,
                                  d_out_num_bits, nullptr, stream);
}

// ---------------------------------------------------------------------------- the whole two-pass build in one C call
// ChannelwisePriorCDFQuantizer.build_entropy_models (quantizer.py:82-150) on one GPU, as vbq_amd.pipeline.EntropyModelBuild enqueues
// it: planes -> pass 1 (solve with raw lengths + bit-length histogram) -> "n + overhead" table -> pass 2 (solve with corrected
// lengths, rank histogram assigned + entropy models looked up in its flush).  Stream-ordered, nothing waits for the host.
extern "C" size_t vbq_build_entropy_models_workspace_bytes(int64_t n_rows, int32_t n_ch, int32_t n_lambda, int32_t N) {
    return vbq_compress_latents_workspace_bytes(n_rows, n_ch, n_lambda, N);      // the same three pieces: planes, index planes, solve
}

extern "C" int vbq_build_entropy_models_f32(const float *d_means_bc, const float *d_spread_bc, int32_t spread_kind, int64_t n_rows,
                                            int32_t n_ch, const float *d_table_lm, const double *h_lambdas, int32_t n_lambda,
                                            int32_t N, const float *d_lut_levels, int64_t n_lut_levels, const float *d_lut_ranks,
                                            int64_t n_lut_ranks, int64_t *d_level_counts, float *d_level_len, float *d_raw_models,
                                            void *d_counts, int32_t counts_are_i32, float *d_models, void *d_workspace,
                                            size_t workspace_bytes, int32_t reserved_workgroups, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows >= 1 && n_ch >= 1 && n_lambda >= 1 && N >= 0 && N <= 15, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_build_entropy_models_f32: bad sizes");
    VBQ_REQUIRE(d_lut_levels && n_lut_levels > n_rows && d_level_counts && d_level_len && d_raw_models && d_counts,
                VBQ_ERR_INVALID_ARGUMENT, "vbq_build_entropy_models_f32: null pointer argument, or a code-length table shorter than n_rows + 1");
    VBQ_REQUIRE((d_models == nullptr) == (d_lut_ranks == nullptr) && (!d_lut_ranks || n_lut_ranks > n_rows), VBQ_ERR_INVALID_ARGUMENT,
                "vbq_build_entropy_models_f32: d_models and d_lut_ranks go together (n_rows + 1 entries)");
    const size_t need = vbq_build_entropy_models_workspace_bytes(n_rows, n_ch, n_lambda, N);
    VBQ_REQUIRE(d_workspace && workspace_bytes >= need, VBQ_ERR_WORKSPACE,
                "vbq_build_entropy_models_f32: workspace of %zu bytes given, %zu needed", workspace_bytes, need);
    VBQ_REQUIRE((reinterpret_cast<uintptr_t>(d_workspace) & 255) == 0, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_build_entropy_models_f32: workspace must be 256-byte aligned");
    const size_t E = (size_t)n_rows * (size_t)n_ch;
    const int64_t N1 = N + 1;
    char *w = static_cast<char *>(d_workspace);
    float *mu_cb = reinterpret_cast<float *>(w);
    w += align256(E * sizeof(float));
    float *sg_cb = reinterpret_cast<float *>(w);
    w += align256(E * sizeof(float));
    uint16_t *idx = reinterpret_cast<uint16_t *>(w);
    w += align256((size_t)n_lambda * E * sizeof(uint16_t));
    const size_t qws = vbq_quantize_workspace_bytes(n_ch, n_lambda, N);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = vbq_prep_planes_f32(d_means_bc, d_spread_bc, spread_kind, n_rows, n_ch, mu_cb, sg_cb, stream);
    if (rc != VBQ_OK) return rc;
    hipError_t e = hipMemsetAsync(d_level_counts, 0, sizeof(int64_t) * (size_t)n_lambda * (size_t)n_ch * (size_t)N1, st);
    if (e != hipSuccess) {
        set_error("vbq_build_entropy_models_f32: hipMemsetAsync: %s", hipGetErrorString(e));
        return VBQ_ERR_LAUNCH;
    }
    // pass 1 (quantizer.py:96-105): raw lengths -> histogram of the winners' bit levels
    rc = vbq_level_counts_f32(mu_cb, sg_cb, n_rows, n_ch, VBQ_LAYOUT_CB, d_table_lm, nullptr, h_lambdas, n_lambda, N, d_level_counts, w, qws,
                              reserved_workgroups, stream);
    if (rc != VBQ_OK) return rc;
    // :105-112, 171-175: -log2 of the smoothed frequencies (tabulated), "n + overhead"
    rc = vbq_code_lengths_from_counts(d_level_counts, 0, (int64_t)n_lambda * n_ch * N1, d_lut_levels, n_lut_levels, (int32_t)N1, d_level_len,
                                      d_raw_models, stream);
    if (rc != VBQ_OK) return rc;
    // pass 2 (:119-146): corrected lengths -> rank indices -> histogram (+ entropy models in its flush)
    rc = vbq_quantize_rows_f32(mu_cb, sg_cb, n_rows, n_ch, VBQ_LAYOUT_CB, d_table_lm, d_level_len, h_lambdas, n_lambda, N, VBQ_MODE_F32, idx,
                               nullptr, nullptr, w, qws, 0, n_rows, 0, reserved_workgroups, stream);
    if (rc != VBQ_OK) return rc;
    return vbq_histogram_models_u16(idx, n_rows, n_ch, n_lambda, N, d_counts, counts_are_i32, d_lut_ranks, n_lut_ranks, d_models, stream);
}

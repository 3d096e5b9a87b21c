// K1, fast form for VBQ_MODE_F32 on channel-uniform workgroups (C == 1 or channel-major
// planes): the hot kernel of the framework.   gfx950 / CDNA4 only.
//
// Same answers as k_quant_flat (vbq_quantize.hip), restructured so that the per-lambda work
// is ~50 VALU ops per element instead of ~130:
//
//  * work in the COST domain: cost_j = fl(d_j + fl(lambda*len_j)) with d_j = fl(0.5*fl(t*t)).
//    Negation is exact in IEEE arithmetic, so fl(-0.5 q - pen) == -fl(0.5 q + pen): the
//    reference's first maximum of the score is the first minimum of the cost.
//  * the two candidates of a bit level share their penalty, and rounding is monotone, so
//    min(cost(L_n), cost(R_n)) = fl(min(dL, dR) + pen_n).  Phase A (once per element) keeps
//    du_n = min(dL_n, dR_n) in registers (11 VGPRs) and parks one packed word per level in an
//    LDS scratch column:  [ top 11 bits of fl(dR - dL), sign flipped | 0:10 | rank of the better side:11 ].
//  * phase B (per lambda): 11 adds of the penalties, a v_min3 tree for the best cost S, then
//    sign(S - c_n) shifted into a bit mask (v_alignbit) -- two ops per level, no VCC traffic --
//    gives the set of levels that attain S.  One LDS read fetches the packed word of the first.
//  * instruction costs measured on gfx950 (tools/ubench.hip): v_add/sub/mul_f32, and/or/xor,
//    add/sub_u32, lshrrev run at 2 cycles per wave64; fma, min/max/min3, alignbit, cmp, cndmask,
//    ffbl, bcnt, lshl_add and ANY op with an SGPR source run at 4.  Hence the penalties are
//    read from an LDS copy into VGPRs (the 44 adds per lambda are then full-rate) instead of
//    being used as SGPR operands.
//  * exactness of the tie rules.  The reference order is [L_0..L_N, R_1..R_N], first maximum
//    wins (utils.py:401).  Two rare events are not decided by the fast path and are flagged:
//      (a) more than one level attains S (a cross-level tie: an L of a deeper level beats an R
//          of a shallower one);
//      (b) the better side of the winning level is R and the two sides are so close that
//          fl(dL + pen) may round onto fl(dR + pen), which would hand the win to L.  The packed
//          gap is a lower bound of dL - dR (>= 2^31 as an unsigned word when L is the better side);
//          the flag is raised when it is below 2^-20 * S (tested as packed word <= bits(2^-19 * S)).
//    A wave with any flagged lane re-solves those lanes for that lambda with the literal
//    21-candidate scan (exact_rank_scan).  Both events have probability ~1e-6 per solve.
#include <stdlib.h>
#include <string.h>

#include "vbq_common.h"

namespace vbq {
namespace {

constexpr int kFastThreads = 256;
#ifndef VBQ_FAST_NE
#define VBQ_FAST_NE 2
#endif
#ifndef VBQ_FAST_WAVES
#define VBQ_FAST_WAVES 4
#endif
#ifndef VBQ_SLOW_INLINE
#define VBQ_SLOW_INLINE __noinline__
#endif
constexpr int kFastNE = VBQ_FAST_NE;      // elements per thread: 4 (16-B loads) or 2 (8-B loads)

// Correctly rounded f32 quotient d / sigma without an f32 division: RN32(RN64(d * RN64(1/sigma))).
// The f64 product is within 2^-52 (relative) of the true quotient, and a quotient of two f32 numbers
// is never closer than 2^-49 (relative) to a midpoint between adjacent f32 values (A*2^24 -
// (2M+1)*B is a non-zero integer for 24-bit A, B), so the final rounding sees the same side of every
// rounding boundary as the exact quotient: the result is bit-identical to IEEE d / sigma.
// v_cvt_f64_f32 + v_mul_f64 + v_cvt_f32_f64 = 12 cycles per wave64 against ~45 for the
// v_div_scale / v_rcp / v_fma x5 / v_div_fmas / v_div_fixup expansion (21 quotients per element).
// tests/test_host_logic.py::test_f64_reciprocal_division_identity checks the identity on 4e7 pairs.
__device__ __forceinline__ float dist_cost(float P, float mu, double rinv) {
    const float d = __fsub_rn(P, mu);
    const float t = (float)__dmul_rn((double)d, rinv);
    const float q = __fmul_rn(t, t);
    return __fmul_rn(0.5f, q);
}

__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }

// Two u16 indices as one 4-byte store.  NT: marked non-temporal -- the indices are written once and read by a later kernel,
// never by this one.  Measured on the Kodak-24 sweep: K1e (32 stores per lane in one burst) 286 -> 261 us; K1 (one store per
// lambda-loop pass) unchanged, and K2 right behind it finds the last planes in the memory-side cache: K1 keeps plain stores.
template <bool NT>
__device__ __forceinline__ void store_idx2(uint32_t *p, uint32_t v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// v_min_f32 / v_max_f32 as they are (IEEE mode: a NaN operand loses).  fminf / fmaxf make the compiler canonicalise
// operands it cannot prove quiet (a v_max x, x in front of every second min of the threshold recurrences).
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmin3abs(float a, float b, float c) {       // min(|a|, |b|, |c|)
    float r;
    asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}


// Minimum of M values as a tree of v_min3_f32 (ceil((M-1)/2) instructions).
template <int M>
__device__ __forceinline__ float min_of(const float (&v)[M]) {
    float t[M];
#pragma unroll
    for (int i = 0; i < M; ++i) t[i] = v[i];
    int m = M;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        if (m <= 1) break;
        int o = 0;
        int i = 0;
#pragma unroll
        for (; i + 2 < m; i += 3) t[o++] = min3f(t[i], t[i + 1], t[i + 2]);
        if (o == 0) { t[o++] = fminf(t[0], t[1]); i = 2; }      // only two values left
        for (; i < m; ++i) t[o++] = t[i];                          // leftovers ride along to the next pass
        m = o;
    }
    return t[0];
}

template <int N>
struct LevelInfo {
    float dL, dR;
    uint32_t posL, posR;
};

// One step of the descent (see vbq_quantize.hip, search_and_score): visits level n of the
// level-major table, returns both neighbours' costs/slots and advances g.
template <int N>
__device__ __forceinline__ LevelInfo<N> descend(const float *tb, int n, float z, double sg, uint32_t &g) {
    const int off = (1 << n) - 1;
    const int m = 1 << n;
    const uint32_t j = g;
    const float pj = tb[off + j];
    const bool below = pj < z;
    LevelInfo<N> r;
    if (n == 0) {
        r.dL = r.dR = dist_cost(pj, z, sg);
        r.posL = r.posR = 0;
    } else {
        int jo = below ? (int)j + 1 : (int)j - 1;
        jo = jo < 0 ? 0 : (jo > m - 1 ? m - 1 : jo);
        bool swap = false;
        if (n == N && below && j == (uint32_t)(m - 1)) { jo = m - 2; swap = true; }
        const float po = tb[off + jo];
        const float dj = dist_cost(pj, z, sg);
        const float dn = dist_cost(po, z, sg);
        const bool j_is_left = below && !swap;
        r.dL = j_is_left ? dj : dn;
        r.dR = j_is_left ? dn : dj;
        r.posL = j_is_left ? j : (uint32_t)jo;
        r.posR = j_is_left ? (uint32_t)jo : j;
    }
    g = 2 * g + (below ? 1u : 0u);
    return r;
}

// Literal restatement of the reference scan for ONE element and ONE lambda (slow path).
template <int N>
__device__ VBQ_SLOW_INLINE uint32_t exact_rank_scan(const float *tb, float z, float sigma, const float *pen) {
    const double sg = __ddiv_rn(1.0, (double)sigma);
    uint32_t g = 0;
    float bestL = 0.f, bestR = 0.f;
    uint32_t rkL = 0, rkR = 0;
    for (int n = 0; n <= N; ++n) {
        const LevelInfo<N> li = descend<N>(tb, n, z, sg, g);
        const float p = pen[n];
        const float cL = __fadd_rn(li.dL, p);
        if (n == 0 || cL < bestL) { bestL = cL; rkL = ((2 * li.posL + 1) << (N - n)) - 1; }
        if (n >= 1) {
            const float cR = __fadd_rn(li.dR, p);
            if (n == 1 || cR < bestR) { bestR = cR; rkR = ((2 * li.posR + 1) << (N - n)) - 1; }
        }
    }
    if (N == 0) return rkL;
    return (bestL <= bestR) ? rkL : rkR;       // every L precedes every R in the reference order
}

// MODE 0: rank indices; 1: + Z_hat / code lengths; 2: no per-element output at all -- the bit LEVEL of every
// winner is counted per (lambda, channel) (the np.bincount(raw_num_bits) of quantizer.py:104), which is all the
// first entropy-model pass needs: no 2 B per solve written, no K2 pass reading them back.  The LDS column that
// parks the packed rank / gap words in modes 0 / 1 holds the counters instead: [L][N+1] x 32 sixteen-bit copies
// (16 words x 2 halves per (lambda, level); lane & 15 picks the word, lane & 16 the half).
template <int N, int MODE>
__global__ void __launch_bounds__(kFastThreads, VBQ_FAST_WAVES)
k_quant_fast(const float *__restrict__ mu, const float *__restrict__ sg, long n_per_ch, long ch_stride, int C,
             const float *__restrict__ table, Lambdas32 lam, const float *__restrict__ len,
             int L, uint16_t *__restrict__ out_idx, float *__restrict__ out_zhat,
             float *__restrict__ out_bits, long E, int vec_ok, int dbg,
             unsigned long long *__restrict__ level_counts) {
    constexpr int T = table_size(N);
    constexpr int N1 = N + 1;
    constexpr int NE = kFastNE;
    constexpr bool EXTRA = MODE == 1;
    constexpr bool COUNT = MODE == 2;
    // dbg (tests only, VBQ_FAST_DEBUG): 1 = send every solve through the literal scan,
    // 2 = never flag (shows that the flags are what keeps the fast path exact)
    const bool never_flag = dbg == 2;
    constexpr int PS = (N1 + 3) & ~3;                 // penalty row, padded to whole 16-byte LDS reads
    __shared__ float tb[T + 1];
    __shared__ uint32_t scratch[N1 * NE * kFastThreads];
    __shared__ __align__(16) float penl[kMaxLambdaChunk * PS];
    const int c = blockIdx.y;
    if (COUNT) {
        for (int i = threadIdx.x; i < N1 * NE * kFastThreads; i += blockDim.x) scratch[i] = 0;
    }
    for (int i = threadIdx.x; i < T; i += blockDim.x) tb[i] = table[(long)c * T + i];
    // A caller-supplied length table may yield a penalty outside {0} U [2^-39, 2^70], the range the tie certificate below
    // assumes: the workgroup that stages such a value sends every solve of its own through the literal scan (same answers,
    // slower) instead of trusting a precondition it cannot see.
    int odd = 0;
    for (int i = threadIdx.x; i < L * PS; i += blockDim.x) {
        const int l = i / PS, n = i - l * PS;
        // pen[l][n] = fl32(lambda_l) * len[l][c][n], len = n for the raw lengths (quantizer.py:167-175, utils.py:394-396)
        const float p = n < N1 ? __fmul_rn(lam.lam[l], len ? len[((long)l * C + c) * N1 + n] : (float)n) : 0.0f;
        penl[i] = p;
        odd |= (len != nullptr && !(p == 0.0f || (p >= 1.8189894e-12f && p <= 1.1805916e21f))) ? 1 : 0;
    }
    const bool force_slow = dbg == 1 || __syncthreads_or(odd) != 0;

    const long base = (long)c * ch_stride;
    const long nquads = (n_per_ch + NE - 1) / NE;

    // (mu, sigma) of element group q.  Channel-last input [n_per_ch][C] (VBQ_LAYOUT_BC_TO_CB): element (row, c) at row * C + c;
    // the lanes of a wave read 4 bytes from 128 different lines, but the 32 channels of a line are read by 32 workgroups in
    // flight together, so the lines come out of L2; the outputs are planes as usual.
    auto load_group = [&](long q, float (&m)[NE], float (&sv)[NE]) {
        const long i0 = q * NE;
        if (!(vec_ok & 2) && (vec_ok & 1) && (i0 + NE <= n_per_ch)) {
            if constexpr (NE == 4) {
                const float4 mv = *reinterpret_cast<const float4 *>(mu + base + i0);
                const float4 s4v = *reinterpret_cast<const float4 *>(sg + base + i0);
                m[0] = mv.x; m[1] = mv.y; m[2] = mv.z; m[3] = mv.w;
                sv[0] = s4v.x; sv[1] = s4v.y; sv[2] = s4v.z; sv[3] = s4v.w;
            } else {
                const float2 mv = *reinterpret_cast<const float2 *>(mu + base + i0);
                const float2 s2v = *reinterpret_cast<const float2 *>(sg + base + i0);
                m[0] = mv.x; m[1] = mv.y;
                sv[0] = s2v.x; sv[1] = s2v.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const bool ok = i0 + k < n_per_ch;
                const long at = (vec_ok & 2) ? (i0 + k) * C + c : base + i0 + k;
                m[k] = ok ? mu[at] : 0.0f;
                sv[k] = ok ? sg[at] : 1.0f;
            }
        }
    };
    const long q0 = (long)blockIdx.x * blockDim.x + threadIdx.x, qstep = (long)gridDim.x * blockDim.x;
    float mn[NE], sn[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) { mn[k] = 0.0f; sn[k] = 1.0f; }
    if (q0 < nquads) load_group(q0, mn, sn);
    const bool resident = (vec_ok & 4) != 0;                // the launcher sized the grid to the resident workgroups
    unsigned int rot = wave_slot();
    for (long q = q0; q < nquads; q += qstep) {
        if (resident) __builtin_amdgcn_s_setprio(3);          // phase A (a chain of LDS round trips) goes first
        const long i0 = q * NE;
        const bool full = (vec_ok & 1) && (i0 + NE <= n_per_ch);
        float m4[NE], s4[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) { m4[k] = mn[k]; s4[k] = sn[k]; }
        // The next group's loads go out BEFORE this group's stores: loads and stores share one in-order counter on gfx9
        // (vmcnt), so a load issued after the last lambda's store could only be waited for together with every store.
        {   // branch-free (the last iteration re-reads its own group): phase A stays one basic block
            const long qn = q + qstep < nquads ? q + qstep : q;
            load_group(qn, mn, sn);
        }

        // ---------------- phase A: descent, per-level best cost + packed side info ----------------
        float du[NE][N1];
        uint32_t g[NE];
        double rinv[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            g[k] = 0;
            rinv[k] = __ddiv_rn(1.0, (double)s4[k]);
        }
        // Slot arithmetic in BYTES (g4 = 4 * slot) so that every LDS address is a register plus an immediate.
        // With G = (number of level-n points below z) = j + [p_n[j] < z], the reference's interval is
        // L = max(G - 1, 0), R = min(G, 2^n - 1) (quantizer.py:75-76 on the padded grid), except on the deepest
        // level, whose grid has no padding: there L stays at 2^N - 2 when z is above the last point.
        const char *tbb = reinterpret_cast<const char *>(tb);
#pragma unroll
        for (int n = 0; n <= N; ++n) {
            constexpr int kWordMask = 0xffe00000u;
            const int off4 = 4 * ((1 << n) - 1);
            const int top4 = 4 * ((1 << n) - 1);                 // byte offset of the last slot of the level
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const float pj = *reinterpret_cast<const float *>(tbb + off4 + g[k]);
                const bool below = pj < m4[k];
                float dL, dR;
                uint32_t lo4, hi4;
                if (n == 0) {
                    dL = dR = dist_cost(pj, m4[k], rinv[k]);
                    lo4 = hi4 = 0;
                } else {
                    const uint32_t G4 = g[k] + (below ? 4u : 0u);
                    int lo = (int)G4 - 4;
                    lo = lo < 0 ? 0 : lo;
                    if (n == N) lo = lo > top4 - 4 ? top4 - 4 : lo;
                    lo4 = (uint32_t)lo;
                    hi4 = G4 > (uint32_t)top4 ? (uint32_t)top4 : G4;
                    dL = dist_cost(*reinterpret_cast<const float *>(tbb + off4 + lo4), m4[k], rinv[k]);
                    dR = dist_cost(*reinterpret_cast<const float *>(tbb + off4 + hi4), m4[k], rinv[k]);
                }
                du[k][n] = vmin(dL, dR);                        // no canonicalising v_max x, x in front (fminf: two per level)
                g[k] = 2 * g[k] + (below ? 4u : 0u);
                if (COUNT) continue;                           // the level alone is wanted: no rank, no gap
                const bool r_better = dR < dL;                  // strict: on equal costs L keeps the level
                const uint32_t b4 = r_better ? hi4 : lo4;
                // rank index ((2 pos + 1) << (N - n)) - 1 of the better side, from 4 * pos
                const uint32_t better = n < N ? (b4 << (N - n - 1)) + ((1u << (N - n)) - 1u) : (b4 >> 1);
                // top 11 bits of fl(dR - dL) with the sign flipped: a (truncated, hence lower) bound of dL - dR
                // when R is the better side, and >= 2^31 -- never "close" -- when L is (dR - dL >= +0)
                const uint32_t gap = __float_as_uint(__fsub_rn(dR, dL)) ^ 0x80000000u;
                scratch[(n * NE + k) * kFastThreads + threadIdx.x] = (gap & kWordMask) | better;
            }
        }

        // ---------------- phase B: one solve per lambda ----------------
        if (resident) {                                       // the VALU-dense part: the four levels in rotation (vbq_common.h);
            set_issue_priority(rot);                            // with three (0..2 under phase A's) two of the four waves always
            rot += 3u;                                          // shared a level and age decided between them: 350 -> 335 us
        }
        for (int l = 0; l < L; ++l) {
            const float *pp = penl + l * PS;
            float p[N1];
#pragma unroll
            for (int n = 0; n < N1; ++n) p[n] = pp[n];
            uint32_t rank[NE];
            // The NE solves are written level-by-level across elements so that neighbouring
            // instructions are independent (one element's chain alone issues ~1 op / 8 cycles).
            float cst[NE][N1];
            float S[NE];
            uint32_t ne[NE];                                    // bit n set <=> cost_n != S
#pragma unroll
            for (int n = 0; n < N1; ++n)
#pragma unroll
                for (int k = 0; k < NE; ++k) cst[k][n] = __fadd_rn(du[k][n], p[n]);
#pragma unroll
            for (int k = 0; k < NE; ++k) S[k] = min_of<N1>(cst[k]);
            {
#pragma unroll
                for (int k = 0; k < NE; ++k) ne[k] = 0;
#pragma unroll
                for (int n = N; n >= 0; --n)
#pragma unroll
                    for (int k = 0; k < NE; ++k)
                        ne[k] = __builtin_amdgcn_alignbit(ne[k], __float_as_uint(__fsub_rn(S[k], cst[k][n])), 31);
            }
            uint64_t fmk[NE];
            if constexpr (COUNT) {
                // level of the winner = shallowest level attaining S, unless several levels attain it (then the
                // reference's L-before-R order decides: literal scan).  Which SIDE wins never changes the level.
                uint64_t any_multi = 0;
                uint32_t lvl[NE];
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    lvl[k] = (uint32_t)__builtin_ctz(~ne[k]);
                    const uint64_t multi = __builtin_amdgcn_uicmp((uint32_t)__popc(ne[k]), (uint32_t)N, 33);
                    fmk[k] = never_flag ? 0ull : (force_slow ? ~0ull : multi);
                    any_multi |= fmk[k];
                }
                if (any_multi != 0) {
                    const uint32_t lane = __lane_id();
#pragma unroll
                    for (int k = 0; k < NE; ++k)
                        if ((fmk[k] >> lane) & 1ull)
                            lvl[k] = (uint32_t)N - (uint32_t)__builtin_ctz(exact_rank_scan<N>(tb, m4[k], s4[k], pp) + 1u);
                }
                const uint32_t lane = threadIdx.x & 63u;
                const uint32_t inc = (lane & 16u) ? 0x10000u : 1u;
#pragma unroll
                for (int k = 0; k < NE; ++k)
                    if (i0 + k < n_per_ch) atomicAdd(&scratch[((uint32_t)(l * N1) + lvl[k]) * 16u + (lane & 15u)], inc);
                continue;
            }
            uint32_t pk[NE];
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int n1 = __builtin_ctz(~ne[k]);           // first (shallowest) level that attains S
                pk[k] = scratch[(n1 * NE + k) * kFastThreads + threadIdx.x];
            }
            // Flags are kept as 64-bit lane masks in SGPRs (v_cmp writes them there): OR-ing them and testing
            // for "any" is scalar work, no VALU op.  LLVM icmp predicates: 33 = ne, 37 = ule.
            uint64_t fm[NE];
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                // more than one level attains S
                const uint64_t multi = __builtin_amdgcn_uicmp((uint32_t)__popc(ne[k]), (uint32_t)N, 33);
                // The gap sits at the float's own bit positions [31:21]: compare bit patterns directly.
                // fl(dL + pen) can meet fl(dR + pen) only if dL - dR <= ulp(S) <= 2^-22 S.  The test
                // (word & 0xffe00000) <= bits(2^-20 S) is implied by word <= bits(2^-19 S) without the mask
                // (doubling a float adds 2^23 to its bit pattern, more than the 21 low bits can hold).
                const uint32_t thr = __float_as_uint(__fmul_rn(S[k], 1.9073486328125e-06f));
                const uint64_t lr_close = __builtin_amdgcn_uicmp(pk[k], thr, 37);
                rank[k] = pk[k] & ((2u << N) - 1u);                // ranks need N + 1 bits, the gap sits above bit 21
                fm[k] = never_flag ? 0ull : (force_slow ? ~0ull : (multi | lr_close));
            }
            uint64_t any_flag = 0;
#pragma unroll
            for (int k = 0; k < NE; ++k) any_flag |= fm[k];
            if (any_flag != 0) {
                const uint32_t lane = __lane_id();
#pragma unroll
                for (int k = 0; k < NE; ++k)
                    if ((fm[k] >> lane) & 1ull) rank[k] = exact_rank_scan<N>(tb, m4[k], s4[k], pp);
            }
            const long o = (long)l * E + base + i0;
            float zh[NE], bt[NE];
            if (EXTRA) {
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    const uint32_t kk = rank[k] + 1;
                    const int tz = __builtin_ctz(kk);
                    const int lvl = N - tz;
                    const uint32_t pos = kk >> (tz + 1);
                    zh[k] = tb[(1 << lvl) - 1 + pos];
                    bt[k] = len ? len[((long)l * C + c) * N1 + lvl] : (float)lvl;
                }
            }
            if (full) {
                if constexpr (NE == 4) {
                    uint2 pk2;
                    pk2.x = rank[0] | (rank[1] << 16);
                    pk2.y = rank[2] | (rank[3] << 16);
                    *reinterpret_cast<uint2 *>(out_idx + o) = pk2;
                    if (EXTRA) {
                        if (out_zhat) *reinterpret_cast<float4 *>(out_zhat + o) = make_float4(zh[0], zh[1], zh[2], zh[3]);
                        if (out_bits) *reinterpret_cast<float4 *>(out_bits + o) = make_float4(bt[0], bt[1], bt[2], bt[3]);
                    }
                } else {
                    store_idx2<false>(reinterpret_cast<uint32_t *>(out_idx + o), rank[0] | (rank[1] << 16));
                    if (EXTRA) {
                        if (out_zhat) *reinterpret_cast<float2 *>(out_zhat + o) = make_float2(zh[0], zh[1]);
                        if (out_bits) *reinterpret_cast<float2 *>(out_bits + o) = make_float2(bt[0], bt[1]);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    if (i0 + k < n_per_ch) {
                        out_idx[o + k] = (uint16_t)rank[k];
                        if (EXTRA) {
                            if (out_zhat) out_zhat[o + k] = zh[k];
                            if (out_bits) out_bits[o + k] = bt[k];
                        }
                    }
                }
            }
        }
    }
    if constexpr (COUNT) {
        __syncthreads();
        for (int i = threadIdx.x; i < L * N1; i += blockDim.x) {
            unsigned int v = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned int w = scratch[i * 16 + j];
                v += (w & 0xffffu) + (w >> 16);
            }
            const int l = i / N1, n = i - l * N1;
            if (v) atomicAdd(&level_counts[((long)l * C + c) * N1 + n], (unsigned long long)v);
        }
    }
}


// =====================================================================================================================
// K1t: the first entropy-model pass (raw lengths, bit levels only) WITHOUT a per-lambda loop.
//
// With the raw lengths of quantizer.py:167-169 the cost of bit level n is a LINE in lambda, du_n + lambda * n, the same
// 11 lines for every lambda of the sweep.  The winner as a function of lambda is therefore a staircase that only steps
// down (the lower envelope of the lines, steeper lines first): it is described by 10 thresholds per element,
//       T_n = sup { lambda : winner(lambda) > n } = max_{j > n} min_{i <= n} (du_i - du_j) / (j - i),   T_0 >= ... >= T_9,
// and the level at lambda is #{ n : lambda < T_n }.  The histogram the pass wants, counts[l][n] = #{elements with level n
// at lambda_l}, follows from the positions a_n = #{ l : lambda_(l) < T_n } of the thresholds in the SORTED sweep:
//       #{ level > n at sorted index l } = #{ elements : a_n > l },
// i.e. from ten 33-bin histograms of a_n -- 10 counter updates per element instead of 32 solves.
//
// Exactness.  The reference does not evaluate real-valued lines: it compares fl(du_n + fl(lambda32 * n)) and breaks ties
// by candidate order.  Each rounded cost is within 2^-23 (1 + 2^-24) of the line's value, relatively; away from every
// threshold the gap between the envelope and any other line is at least the distance |lambda - T| (slopes differ by >= 1),
// so the rounded comparison and the real one agree whenever that distance exceeds 2^-22 of the envelope's height, which
// level n's own line bounds from above.  Thresholds computed in f32 carry a relative error below 2^-22 (max / min commute
// with the monotone map x -> x (1 +- d)).  Hence: the two sweep points next to every threshold are tested against the
// guard band  2^-20 (du_n + |T_n| (n + 1));  a sweep point inside a band -- or any non-finite cost -- is re-solved for
// that element with the literal 21-candidate scan and the histogram corrected by (+1 exact level, -1 predicted level).
// About 1 threshold in 10 000 takes that path (Kodak-24 sweep: 9 010 of 3.0e8 (element, lambda) pairs, 8 levels changed).  tests/test_gpu_twopass.py forces it (exact ties, equal
// costs, thresholds on sweep points) and tools/stress_parity.py --levels compares 1e9+ solves with the C oracle.
#ifndef VBQ_HULL_WAVES
#define VBQ_HULL_WAVES 4
#endif
#ifndef VBQ_HULL_NE
#define VBQ_HULL_NE 2
#endif
#ifndef VBQ_HULL_THREADS
#define VBQ_HULL_THREADS 256
#endif
constexpr int kHullThreads = VBQ_HULL_THREADS;
#ifndef VBQ_HULL_COPIES
#define VBQ_HULL_COPIES 16
#endif
constexpr int kHullKeys = 2048;         // 16 octaves of 128 buckets
constexpr int kFixQueue = 192;          // K1t: deferred fix-ups per workgroup (about ten are expected per 2 300 elements)
struct HullSweep {
    float lam[32];          // the sweep rounded to f32, ascending
    unsigned char perm[32]; // position of lam[l] in the caller's order
    int L, key0, nkeys;     // keys = float bits >> 16 (sign, exponent, 7 mantissa bits); bucket b <-> key0 + b
    unsigned char lut[kHullKeys];   // lut[b] = #{ l : lam[l] below the lower edge of bucket b }
};
constexpr float kHullBig = 3.0e38f;

// Distortion of the NEARER neighbour of z on every bit level (du is monotone in |P - z|, so one exact quotient per level).
// The visited point is one neighbour of z on its level; the other one sits on z's side of it (the same point again at the
// rim: the deepest level's one-slot-back quirk, quantizer.py:54-57, only ever moves the FARTHER candidate, which a minimum
// does not see).
template <int N>
__device__ __forceinline__ void hull_du_nearest(const char *tbb, float z, float sigma, float (&du)[N + 1]) {
    uint32_t g = 0;
    const double rinv = __ddiv_rn(1.0, (double)sigma);
#pragma unroll
    for (int n = 0; n <= N; ++n) {
        const int off4 = 4 * ((1 << n) - 1);
        const int top4 = off4;
        const float pj = *reinterpret_cast<const float *>(tbb + off4 + g);
        const bool below = pj < z;
        float dmin;
        if (n == 0) {
            dmin = __fsub_rn(pj, z);
        } else {
            int o4 = (int)g + (below ? 4 : -4);
            o4 = o4 < 0 ? 0 : (o4 > top4 ? top4 : o4);
            const float po = *reinterpret_cast<const float *>(tbb + off4 + o4);
            dmin = fminf(fabsf(__fsub_rn(pj, z)), fabsf(__fsub_rn(po, z)));
        }
        const float t = (float)__dmul_rn((double)dmin, rinv);
        du[n] = __fmul_rn(0.5f, __fmul_rn(t, t));
        g = 2 * g + (below ? 4u : 0u);
    }
}

// The ten thresholds T_n = max_{j > n} min_{i <= n} (du_i - du_j) / (j - i) of an element, clamped into the tables' range
// (inf / NaN only come from non-finite costs, which are flagged separately).
template <int N>
__device__ __forceinline__ void hull_thresholds(const float (&du)[N + 1], float (&Tn)[N]) {
    constexpr int N1 = N + 1;
    float Pm[N1];                                              // Pm[j] = min_{i <= n} (du_i - du_j) / (j - i)
#pragma unroll
    for (int n = 0; n < N; ++n) {
#pragma unroll
        for (int j = n + 1; j < N1; ++j) {
            const float r = __fmul_rn(__fsub_rn(du[n], du[j]), 1.0f / (float)(j - n));
            Pm[j] = n == 0 ? r : vmin(Pm[j], r);
        }
        float t = Pm[n + 1];
        int j = n + 2;
#pragma unroll
        for (; j + 1 < N1; j += 2) t = vmax3(t, Pm[j], Pm[j + 1]);
        if (j < N1) t = vmax(t, Pm[j]);
        Tn[n] = vmin(t, 1.0e38f);
    }
}

template <int N>
__device__ __forceinline__ float hull_max_du(const float (&du)[N + 1]) {
    constexpr int N1 = N + 1;
    float big = du[0];
#pragma unroll
    for (int j = 1; j + 1 < N1; j += 2) big = vmax3(big, du[j], du[j + 1]);
    if ((N1 & 1) == 0) big = vmax(big, du[N1 - 1]);
    return big;
}

// Position of a threshold in the sorted sweep, a = #{ l : lam_(l) < T }: bucket of the threshold's bit pattern (an
// arithmetic shift keeps T <= 0 negative, so one median clamps "below the sweep", "above it" and the table range), then the
// one sweep point that may share the bucket.  nb = { lam_(cnt-1), lam_(cnt), lam_(cnt+1) }: T lies between the outer two,
// so its distance to the sweep is the smallest of the three distances.
__device__ __forceinline__ uint32_t hull_position(float t, int key0, int nkeys, const unsigned char *lut, const float4 *rec,
                                                  float4 &nb) {
    const int key = min(max(((int)__float_as_uint(t) >> 16) - key0, 0), nkeys - 1);    // v_med3_i32
    const uint32_t cnt = lut[key];
    nb = rec[cnt];
    return cnt + (nb.y < t ? 1u : 0u);                        // lam_(a-1) < T <= lam_(a)
}

// Guard band |lambda - T_n| <= 2^-20 (du_n + |T_n| (n + 1)): every sweep point outside it is decided by the lines.
__device__ __forceinline__ float hull_band(float t, int n, float du_n) {
    return __fmul_rn(fmaf(fabsf(t), (float)(n + 1), du_n), 9.5367431640625e-07f);
}

// Bit level of the winner at one sweep point from the per-level best distortions (the counting kernels' question: which
// SIDE wins never changes the level).  multi: several levels attain the minimum, the reference's candidate order decides.
template <int N>
__device__ __forceinline__ uint32_t level_of_min(const float (&du)[N + 1], const float *pen, bool &multi) {
    constexpr int N1 = N + 1;
    float cst[N1];
#pragma unroll
    for (int n = 0; n < N1; ++n) cst[n] = __fadd_rn(du[n], pen[n]);
    const float S = min_of<N1>(cst);
    uint32_t ne = 0;                                           // bit n set <=> cost_n != S
#pragma unroll
    for (int n = N; n >= 0; --n) ne = __builtin_amdgcn_alignbit(ne, __float_as_uint(__fsub_rn(S, cst[n])), 31);
    multi = (uint32_t)__popc(ne) != (uint32_t)N;
    return (uint32_t)__builtin_ctz(~ne);
}

// Rare path of K1t for element k of the lanes in `lanes` (a sweep point inside a guard band, a non-finite cost, or the test
// switch that sends everything here), entered once per iteration AFTER the straight block of both elements: thresholds
// and positions again from the distortions still in registers -- exactly as the hot block formed them --, then every
// sweep point inside a band is re-solved and the histogram corrected by (+1 exact level, -1 predicted level).
template <int N>
__device__ __forceinline__ void hull_counts_fix(uint64_t lanes, const float (&du_hot)[N + 1], const float *tb, float z, float sigma,
                                                int L, int key0, int nkeys, const unsigned char *perm, const unsigned char *lut,
                                                const float4 *rec, const float *penl, int *corr, bool all_points) {
    constexpr int N1 = N + 1;
    constexpr int PS = (N1 + 3) & ~3;
    const bool mine = (lanes >> (threadIdx.x & 63u)) & 1ull;
    float du[N1];
#pragma unroll
    for (int n = 0; n < N1; ++n) {                             // opaque copies: nothing below may be hoisted into the hot block
        float d = du_hot[n];
        asm volatile("" : "+v"(d));
        du[n] = d;
    }
    float Tn[N];
    uint32_t apos[N];
    hull_thresholds<N>(du, Tn);
    uint32_t fl = (!(hull_max_du<N>(du) < kHullBig) || all_points) ? 0xffffffffu : 0u;   // non-finite costs: every lambda re-solved
    float G[N];
    uint32_t near = 0;                                         // bit n: a sweep point lies inside the band of threshold n
#pragma unroll
    for (int n = 0; n < N; ++n) {                              // straight code: the twenty table reads overlap
        float4 nb;
        apos[n] = hull_position(Tn[n], key0, nkeys, lut, rec, nb);
        G[n] = hull_band(Tn[n], n, du[n]);
        const float dist = vmin3abs(__fsub_rn(Tn[n], nb.x), __fsub_rn(Tn[n], nb.y), __fsub_rn(Tn[n], nb.z));
        near |= (mine && dist <= G[n]) ? (1u << n) : 0u;
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        if ((near >> n) & 1u) {
            // list the sweep points inside this band: they are consecutive and next to T, so walk outwards from its
            // position (lam_(a-1) < T <= lam_(a); rec[l + 1].x = lam_(l)) -- one or two steps, not a pass over the sweep
            for (int l = (int)apos[n] - 1; l >= 0 && fabsf(__fsub_rn(rec[l + 1].x, Tn[n])) <= G[n]; --l) fl |= 1u << l;
            for (int l = (int)apos[n]; l < L && fabsf(__fsub_rn(rec[l + 1].x, Tn[n])) <= G[n]; ++l) fl |= 1u << l;
        }
    }
    fl &= L >= 32 ? 0xffffffffu : ((1u << L) - 1u);
    if (!mine) fl = 0u;
    while (fl != 0u) {
        const int l = __builtin_ctz(fl);
        fl &= fl - 1u;
        const int lo_ = perm[l];
        bool multi;
        int n_ex = (int)level_of_min<N>(du, penl + lo_ * PS, multi);
        if (multi) n_ex = N - __builtin_ctz(exact_rank_scan<N>(tb, z, sigma, penl + lo_ * PS) + 1u);
        int n_pred = 0;                                        // the level the counters assumed: #{ n : a_n > l }
#pragma unroll
        for (int n = 0; n < N; ++n) n_pred += apos[n] > (uint32_t)l ? 1 : 0;
        if (n_ex != n_pred) {
            atomicAdd(&corr[l * N1 + n_ex], 1);
            atomicSub(&corr[l * N1 + n_pred], 1);
        }
    }
}

template <int N>
__global__ void __launch_bounds__(kHullThreads, (VBQ_HULL_WAVES * 256) / kHullThreads)
k_level_counts_hull(const float *__restrict__ mu, const float *__restrict__ sg, long n_per_ch, long ch_stride, int C,
                    const float *__restrict__ table, Lambdas32 lam, HullSweep sw, int vec_ok,
                    unsigned long long *__restrict__ level_counts, int dbg) {
    constexpr int T = table_size(N);
    constexpr int N1 = N + 1;
    constexpr int NE = VBQ_HULL_NE;
    constexpr int KC = VBQ_HULL_COPIES;                      // words per (level, position) counter; two 16-bit halves each
    constexpr int PS = (N1 + 3) & ~3;
    constexpr int LB = 33;                                    // positions 0..32
    __shared__ float tb[T + 1];
    __shared__ __align__(16) float penl[kMaxLambdaChunk * PS];
    __shared__ __align__(4) unsigned char lut[kHullKeys];
    __shared__ float4 rec[34];                                // rec[i] = { lam[i-1], lam[i], lam[i+1], - } with -big / +big outside
    __shared__ unsigned int H[N * LB * KC];                   // [n][a][16 words x 2 halves]
    __shared__ int corr[kMaxLambdaChunk * N1];
    __shared__ unsigned int n_valid;
    __shared__ unsigned char perm_s[32];
    __shared__ unsigned int fixq[kFixQueue], fixn;             // elements whose fix-up waits for the end of the loop
    const int c = blockIdx.y;
    const int L = sw.L;
    for (int i = threadIdx.x; i < T; i += blockDim.x) tb[i] = table[(long)c * T + i];
    for (int i = threadIdx.x; i < L * PS; i += blockDim.x) {
        const int l = i / PS, n = i - l * PS;
        penl[i] = n < N1 ? __fmul_rn(lam.lam[l], (float)n) : 0.0f;       // raw lengths, the caller's lambda order
    }
    if (threadIdx.x < 32) perm_s[threadIdx.x] = sw.perm[threadIdx.x];
    for (int i = threadIdx.x; i < N * LB * KC; i += blockDim.x) H[i] = 0;
    for (int i = threadIdx.x; i < L * N1; i += blockDim.x) corr[i] = 0;
    for (int b = threadIdx.x; b < kHullKeys / 4; b += blockDim.x)   // the bucket table travels in the kernel arguments
        reinterpret_cast<uint32_t *>(lut)[b] = reinterpret_cast<const uint32_t *>(sw.lut)[b];
    if (threadIdx.x < 34) {
        const int i = (int)threadIdx.x;
        auto at = [&](int l) { return l < 0 ? -kHullBig : (l < L ? sw.lam[l < 32 ? l : 31] : kHullBig); };
        rec[i] = make_float4(at(i - 1), at(i), at(i + 1), 0.0f);
    }
    if (threadIdx.x == 0) { n_valid = 0; fixn = 0; }
    __syncthreads();

    const bool force_slow = dbg == 1, never_flag = dbg == 2;
    const long base = (long)c * ch_stride;
    const long nquads = (n_per_ch + NE - 1) / NE;
    const char *tbb = reinterpret_cast<const char *>(tb);
    const unsigned int lane = threadIdx.x & 63u;
    const unsigned int inc = (lane & (unsigned)KC) ? 0x10000u : 1u;
    const unsigned int copy = lane & (unsigned)(KC - 1);
    const int key0 = sw.key0, nkeys = sw.nkeys;
    unsigned int my_valid = 0;

    const long q0 = (long)blockIdx.x * blockDim.x + threadIdx.x, qstep = (long)gridDim.x * blockDim.x;
    int it = 0;
    unsigned int rot = wave_slot();
    for (long q = q0; q < nquads; q += qstep, ++it) {
        rotate_issue_priority(rot);                            // resident grid: see vbq_common.h
        const long i0 = q * NE;
        const bool full = (vec_ok & 1) && (i0 + NE <= n_per_ch);
        float m4[NE], s4[NE];
        if (vec_ok & 2) {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const bool ok = i0 + k < n_per_ch;
                m4[k] = ok ? mu[(i0 + k) * C + c] : 0.0f;
                s4[k] = ok ? sg[(i0 + k) * C + c] : 1.0f;
            }
        } else if (full && NE == 2) {
            const float2 mv = *reinterpret_cast<const float2 *>(mu + base + i0);
            const float2 sv = *reinterpret_cast<const float2 *>(sg + base + i0);
            m4[0] = mv.x; m4[NE - 1] = mv.y;
            s4[0] = sv.x; s4[NE - 1] = sv.y;
        } else {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const bool ok = i0 + k < n_per_ch;
                m4[k] = ok ? mu[base + i0 + k] : 0.0f;
                s4[k] = ok ? sg[base + i0 + k] : 1.0f;
            }
        }
        // The whole iteration is ONE straight block for both elements (their table reads and LDS round trips overlap); the
        // only branch, taken by about one wave in ten, goes to the out-of-line fix-up at the end.
        float du[NE][N1];
#pragma unroll
        for (int k = 0; k < NE; ++k) hull_du_nearest<N>(tbb, m4[k], s4[k], du[k]);
        uint64_t fix[NE];                                      // lanes whose element k needs the fix-up
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const bool valid = i0 + k < n_per_ch;
            my_valid += valid ? 1u : 0u;
            const unsigned int vinc = valid ? inc : 0u;        // padding lanes add 0: no exec juggling
            float Tn[N];
            hull_thresholds<N>(du[k], Tn);
            fix[k] = __builtin_amdgcn_ballot_w64(!(hull_max_du<N>(du[k]) < kHullBig)) | (force_slow ? ~0ull : 0ull);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                float4 nb;
                const uint32_t a = hull_position(Tn[n], key0, nkeys, lut, rec, nb);
                const float G = hull_band(Tn[n], n, du[k][n]);
                const float dist = vmin3abs(__fsub_rn(Tn[n], nb.x), __fsub_rn(Tn[n], nb.y), __fsub_rn(Tn[n], nb.z));
                fix[k] |= __builtin_amdgcn_ballot_w64(dist <= G);
                atomicAdd(&H[((uint32_t)n * LB + a) * (unsigned)KC + copy], vinc);
            }
            fix[k] &= __builtin_amdgcn_ballot_w64(valid);      // lane masks are scalars: no branch inside the block
        }
        uint64_t any_fix = 0;
#pragma unroll
        for (int k = 0; k < NE; ++k) any_fix |= fix[k];
        if (any_fix != 0 && !never_flag) {
            // About one iteration in ten has such a lane, and the fix-up runs for the whole wave: the element is only NOTED
            // (iteration, thread, k) in a workgroup queue and repaired after the loop, where all the queued elements of
            // the workgroup share one pass.  A full queue (adversarial data, the test switch) repairs on the spot.
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const bool mine = (fix[k] >> lane) & 1ull;
                unsigned int slot = kFixQueue;
                if (mine) slot = atomicAdd(&fixn, 1u);
                if (mine && slot < (unsigned)kFixQueue) fixq[slot] = ((unsigned int)it << 9) | (threadIdx.x << 1) | (unsigned)k;
                const uint64_t now = __builtin_amdgcn_ballot_w64(mine && slot >= (unsigned)kFixQueue);
                if (now != 0ull)
                    hull_counts_fix<N>(now, du[k], tb, m4[k], s4[k], L, key0, nkeys, perm_s, lut, rec, penl, corr, force_slow);
            }
        }
    }
    __syncthreads();
    {   // the queued elements: (mu, sigma) again, the distortions again, then the same fix-up, one element per lane
        const unsigned int nq = fixn < (unsigned)kFixQueue ? fixn : (unsigned)kFixQueue;
        for (unsigned int i = threadIdx.x; i < ((nq + 63u) & ~63u); i += blockDim.x) {
            const bool on = i < nq;
            const unsigned int e = on ? fixq[i] : 0u;
            const long qq = q0 + (long)(e >> 9) * qstep + (long)((e >> 1) & 255u) - (long)threadIdx.x;
            const long el = qq * NE + (long)(e & 1u);
            const long at = (vec_ok & 2) ? el * C + c : base + el;
            const float z = on ? mu[at] : 0.0f, sgm = on ? sg[at] : 1.0f;
            float du[N1];
            hull_du_nearest<N>(tbb, z, sgm, du);
            const uint64_t lanes = __builtin_amdgcn_ballot_w64(on);
            hull_counts_fix<N>(lanes, du, tb, z, sgm, L, key0, nkeys, perm_s, lut, rec, penl, corr, force_slow);
        }
    }
    atomicAdd(&n_valid, my_valid);
    __syncthreads();
    // counts[l][n] = #{a_{n-1} > l} - #{a_n > l} + corrections, with #{a_{-1} > l} = all elements and #{a_N > l} = 0
    unsigned int *hs = reinterpret_cast<unsigned int *>(penl);          // [N][LB] sums over the 32 partial counters (penl is done)
    static_assert(N * LB <= kMaxLambdaChunk * PS, "hs must fit the penalty staging area");
    for (int i = threadIdx.x; i < N * LB; i += blockDim.x) {
        unsigned int v = 0;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const unsigned int w = H[i * KC + j];
            v += (w & 0xffffu) + (w >> 16);
        }
        hs[i] = v;
    }
    __syncthreads();
    if (threadIdx.x < N) {                                              // suffix sums: hs[n][a] := #{a_n >= a}
        unsigned int run = 0;
        for (int a = LB - 1; a >= 0; --a) { run += hs[threadIdx.x * LB + a]; hs[threadIdx.x * LB + a] = run; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * N1; i += blockDim.x) {
        const int l = i / N1, n = i - l * N1;
        const long hi_cnt = n == 0 ? (long)n_valid : (long)hs[(n - 1) * LB + l + 1];
        const long lo_cnt = n == N ? 0 : (long)hs[n * LB + l + 1];
        const long v = hi_cnt - lo_cnt + corr[i];
        if (v) atomicAdd(&level_counts[((long)sw.perm[l] * C + c) * N1 + n], (unsigned long long)v);
    }
}


// =====================================================================================================================
// K1e: rank INDICES for a whole lambda sweep with the raw lengths of quantizer.py:167-169, from K1t's thresholds.
//
// Same lines, same thresholds, same positions and guard bands as K1t; what changes is the product: not ten counter updates
// per element but one index per (element, lambda).  The better side's rank of every level goes to a per-lane LDS column
// rk[N - level]; every position a_n adds 1 to a 4-bit field "levels lost at sorted sweep point a_n" of a per-lane column
// (private: no contention); the sweep is then emitted by walking the column.  The fields of a word sum to at most 10, so
// one multiplication by 0x11111111 turns eight of them into their running sums, a shift and a v_and_or_b32 turn a sum
// into the LDS address of rk[level], and two elements leave as one 4-byte store per lambda: ~9 instructions per lambda for
// two elements, against ~95 in k_quant_fast.  (The notebook's f64 variant of the same idea is k_quant_notebook_hull.)
//
// Exactness on top of K1t's bands: WHICH SIDE of the winning level.  The reference scans [L_0..L_N, R_1..R_N] and keeps
// the first maximum, so inside a level R wins only if fl(dR + pen) < fl(dL + pen); with dR < dL that can fail when
// dL - dR is below an ulp of the cost.  A level n >= 1 only wins where its cost is at most level n-1's, i.e. with a cost
// <= du_n + n du_(n-1); a right side that is better by less than 2^-19 of that marks the level, and the sweep points at
// which a marked level wins (sorted positions a_n .. a_(n-1) - 1) are re-solved with the literal scan like the points
// inside a guard band.  Ties between levels only occur inside bands (K1t's argument), where the literal scan applies the
// reference's candidate order.
#ifndef VBQ_K1E_STAGE
#define VBQ_K1E_STAGE 5
#endif
#ifndef VBQ_K1E_WAVES
#define VBQ_K1E_WAVES 4
#endif
// LFIX = 32 / 16: the sweep has exactly that many points (four / two full words of the "levels lost" column): the emission is one straight block
// -- all 64 rank reads of a lane pair in flight together -- instead of four blocks with a branch and a full LDS latency each.
template <int N, int LFIX, bool BUF>
__global__ void __launch_bounds__(256, VBQ_K1E_WAVES)
k_quant_hull_idx(const float *__restrict__ mu, const float *__restrict__ sg, long n_per_ch, long ch_stride, int C,
                 const float *__restrict__ table, Lambdas32 lam, HullSweep sw, int vec_ok,
                 uint16_t *__restrict__ out_idx, long E, int dbg) {
    constexpr int T = table_size(N);
    constexpr int N1 = N + 1;
    constexpr int NE = 2;
    constexpr int PS = (N1 + 3) & ~3;
    constexpr int CW = 5;                                     // words of eight 4-bit fields: sorted positions 0 .. 32
    struct Lds {
        unsigned short rk[N1 * NE * 256];                     // rank of the better side, [N - level][thread][element]: a lane's two
                                                              // ranks share a dword, so the 32 lanes of an LDS group sit on 32 banks
        float tb[T + 1];
        float penl[kMaxLambdaChunk * PS];                     // fl32(lambda) * n in the caller's order (literal scan only)
        unsigned char lut[kHullKeys];
        float4 rec[34];                                       // rec[i] = { lam[i-1], lam[i], lam[i+1], - }, -big / +big outside
        unsigned char perm_s[32];
        unsigned int cnt[CW * NE * 256];                      // [word][element][thread]: levels lost at sorted sweep point l
    };
    __shared__ __align__(16) Lds lds;
    unsigned short *rk = lds.rk;
    float *tb = lds.tb, *penl = lds.penl;
    unsigned char *lut = lds.lut, *perm_s = lds.perm_s;
    float4 *rec = lds.rec;
    unsigned int *cnt = lds.cnt;
    const int c = blockIdx.y;
    const int L = LFIX ? LFIX : sw.L;
    const unsigned int tid = threadIdx.x;
    for (int i = tid; i < T; i += blockDim.x) tb[i] = table[(long)c * T + i];
    for (int i = tid; i < L * PS; i += blockDim.x) {
        const int l = i / PS, n = i - l * PS;
        penl[i] = n < N1 ? __fmul_rn(lam.lam[l], (float)n) : 0.0f;
    }
    for (int b = tid; b < kHullKeys / 4; b += blockDim.x)
        reinterpret_cast<uint32_t *>(lut)[b] = reinterpret_cast<const uint32_t *>(sw.lut)[b];
    if (tid < 34) {
        const int i = (int)tid;
        auto at = [&](int l) { return l < 0 ? -kHullBig : (l < L ? sw.lam[l < 32 ? l : 31] : kHullBig); };
        rec[i] = make_float4(at(i - 1), at(i), at(i + 1), 0.0f);
    }
    if (tid < 32) perm_s[tid] = sw.perm[tid];
    for (int i = tid; i < CW * NE * 256; i += blockDim.x) cnt[i] = 0;
    __syncthreads();

    const bool force_slow = dbg == 1, never_flag = dbg == 2;
    const uint32_t all_l = L >= 32 ? 0xffffffffu : ((1u << L) - 1u);
    const long base = (long)c * ch_stride;
    const long npairs = (n_per_ch + NE - 1) / NE;
    const char *tbb = reinterpret_cast<const char *>(tb);
    const unsigned int lane = tid & 63u;

    // BUF: every index plane starts within 4 GB of out_idx (the host checks): stores address them through one buffer descriptor
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out_idx, 0, -1 /* 4 GB */, 0x00020000);
    const uint32_t plane_bytes = (uint32_t)(2 * E);
    unsigned int rot = wave_slot();
    for (long q = (long)blockIdx.x * blockDim.x + tid; q < npairs; q += (long)gridDim.x * blockDim.x) {
        rotate_issue_priority(rot);                            // resident grid: see vbq_common.h
        const long i0 = q * NE;
        const bool full = (vec_ok & 1) && (i0 + NE <= n_per_ch);
        float m4[NE], s4[NE];
        if (vec_ok & 2) {                                      // channel-last input (VBQ_LAYOUT_BC_TO_CB), see k_quant_fast
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const bool ok = i0 + k < n_per_ch;
                m4[k] = ok ? mu[(i0 + k) * C + c] : 0.0f;
                s4[k] = ok ? sg[(i0 + k) * C + c] : 1.0f;
            }
        } else if (full) {
            const float2 mv = *reinterpret_cast<const float2 *>(mu + base + i0);
            const float2 sv = *reinterpret_cast<const float2 *>(sg + base + i0);
            m4[0] = mv.x; m4[1] = mv.y;
            s4[0] = sv.x; s4[1] = sv.y;
        } else {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const bool ok = i0 + k < n_per_ch;
                m4[k] = ok ? mu[base + i0 + k] : 0.0f;
                s4[k] = ok ? sg[base + i0 + k] : 1.0f;
            }
        }
        // ---- phase A (as k_quant_fast): both neighbours' distortions on every level, the better side's rank
        float du[NE][N1];
        uint32_t tiny[NE] = {0, 0};                            // bit n: R is the better side of level n by a hair
        {
            uint32_t g[NE] = {0, 0};
            double rinv[NE];
#pragma unroll
            for (int k = 0; k < NE; ++k) rinv[k] = __ddiv_rn(1.0, (double)s4[k]);
#pragma unroll
            for (int n = 0; n <= N; ++n) {
                const int off4 = 4 * ((1 << n) - 1);
                const int top4 = off4;                         // byte offset of the last slot of the level
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    const float pj = *reinterpret_cast<const float *>(tbb + off4 + g[k]);
                    const bool below = pj < m4[k];
                    float dL, dR;
                    uint32_t lo4, hi4;
                    if (n == 0) {
                        dL = dR = dist_cost(pj, m4[k], rinv[k]);
                        lo4 = hi4 = 0;
                    } else {
                        const uint32_t G4 = g[k] + (below ? 4u : 0u);
                        int lo = (int)G4 - 4;
                        lo = lo < 0 ? 0 : lo;
                        if (n == N) lo = lo > top4 - 4 ? top4 - 4 : lo;       // the deepest level's one-slot-back quirk
                        lo4 = (uint32_t)lo;
                        hi4 = G4 > (uint32_t)top4 ? (uint32_t)top4 : G4;
                        dL = dist_cost(*reinterpret_cast<const float *>(tbb + off4 + lo4), m4[k], rinv[k]);
                        dR = dist_cost(*reinterpret_cast<const float *>(tbb + off4 + hi4), m4[k], rinv[k]);
                    }
                    const bool r_better = dR < dL;              // strict: on equal distortions L keeps the level
                    du[k][n] = r_better ? dR : dL;
                    g[k] = 2 * g[k] + (below ? 4u : 0u);
                    const uint32_t b4 = r_better ? hi4 : lo4;
                    const uint32_t better = n < N ? (b4 << (N - n - 1)) + ((1u << (N - n)) - 1u) : (b4 >> 1);
                    rk[(((N - n) * 256 + tid) * NE) + k] = (unsigned short)better;
                    if (n >= 1) {
                        const float bound = __fmul_rn(fmaf((float)n, du[k][n - 1], du[k][n]), 1.9073486328125e-06f);
                        tiny[k] |= (r_better && !(__fsub_rn(dL, dR) > bound)) ? (1u << n) : 0u;
                    }
                }
            }
        }
        // ---- thresholds, positions in the sorted sweep, guard bands (K1t), the "levels lost" column
        uint32_t flags[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const bool valid = i0 + k < n_per_ch;
            float Pm[N1], Tn[N];
            uint64_t near[N];
            float big = du[k][0];
#pragma unroll
            for (int j = 1; j + 1 < N1; j += 2) big = vmax3(big, du[k][j], du[k][j + 1]);
            if ((N1 & 1) == 0) big = vmax(big, du[k][N1 - 1]);
            uint32_t fl = (!(big < kHullBig) || force_slow) ? 0xffffffffu : 0u;
            uint64_t any_near = 0;
#pragma unroll
            for (int n = 0; n < N; ++n) {
#pragma unroll
                for (int j = n + 1; j < N1; ++j) {
                    const float r = __fmul_rn(__fsub_rn(du[k][n], du[k][j]), 1.0f / (float)(j - n));
                    Pm[j] = n == 0 ? r : vmin(Pm[j], r);
                }
                float t = Pm[n + 1];
                {
                    int j = n + 2;
#pragma unroll
                    for (; j + 1 < N1; j += 2) t = vmax3(t, Pm[j], Pm[j + 1]);
                    if (j < N1) t = vmax(t, Pm[j]);
                }
                Tn[n] = vmin(t, 1.0e38f);
            }
            uint32_t c0[N];
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int key = min(max(((int)__float_as_uint(Tn[n]) >> 16) - sw.key0, 0), sw.nkeys - 1);
                c0[n] = lut[key];
            }
#pragma unroll
            for (int h = 0; h < N; h += VBQ_K1E_STAGE) {
                float4 nb[VBQ_K1E_STAGE];
#pragma unroll
                for (int i = 0; i < VBQ_K1E_STAGE; ++i) nb[i] = rec[c0[h + i]];
#pragma unroll
                for (int i = 0; i < VBQ_K1E_STAGE; ++i) {
                    const int n = h + i;
                    const float t = Tn[n];
                    const uint32_t a = c0[n] + (nb[i].y < t ? 1u : 0u);          // lam_(a-1) < T <= lam_(a)
                    const float G = __fmul_rn(fmaf(fabsf(t), (float)(n + 1), du[k][n]), 9.5367431640625e-07f);
                    const float dist = vmin3abs(__fsub_rn(t, nb[i].x), __fsub_rn(t, nb[i].y), __fsub_rn(t, nb[i].z));
                    near[n] = __builtin_amdgcn_ballot_w64(dist <= G);
                    any_near |= near[n];
                    atomicAdd(&cnt[((a >> 3) * NE + k) * 256 + tid], 1u << (4u * (a & 7u)));
                }
            }
            if (any_near != 0) {                               // rare: list the sweep points inside the band(s)
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    if (near[n] == 0) continue;
                    if ((near[n] >> lane) & 1ull) {
                        const float G = __fmul_rn(fmaf(fabsf(Tn[n]), (float)(n + 1), du[k][n]), 9.5367431640625e-07f);
                        for (int l = 0; l < L; ++l)
                            fl |= (fabsf(__fsub_rn(sw.lam[l], Tn[n])) <= G) ? (1u << l) : 0u;
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(tiny[k] != 0u) != 0ull) {      // rare: a marked level and where it wins
                auto pos_of = [&](float t) {                   // the position of a threshold again (not kept: registers)
                    const int key = min(max(((int)__float_as_uint(t) >> 16) - sw.key0, 0), sw.nkeys - 1);
                    const uint32_t cc = lut[key];
                    return cc + (rec[cc].y < t ? 1u : 0u);
                };
#pragma unroll
                for (int n = 1; n <= N; ++n) {
                    if ((tiny[k] >> n) & 1u) {
                        const uint64_t hi = (1ull << pos_of(Tn[n - 1])) - 1ull;       // sorted points below a_(n-1)
                        const uint64_t lo = n < N ? (1ull << pos_of(Tn[n < N ? n : 0])) - 1ull : 0ull;
                        fl |= (uint32_t)(hi & ~lo);
                    }
                }
            }
            flags[k] = (valid && !never_flag) ? (fl & all_l) : 0u;
        }
        // ---- emit the sweep (see k_quant_notebook_hull): level index r = N - level only grows along the sorted sweep
        uint32_t cw[CW][NE];
#pragma unroll
        for (int wd = 0; wd < CW; ++wd)
            if (wd * 8 <= L) {
#pragma unroll
                for (int k = 0; k < NE; ++k) cw[wd][k] = atomicExch(&cnt[(wd * NE + k) * 256 + tid], 0u);    // read and clear
            }
        uint16_t *oi = out_idx + base + i0;
        const uint32_t voff2 = (uint32_t)(2 * (base + i0));    // byte offset inside a plane (planes hold fewer than 2^31 elements)
        if (full) {
            const char *rkb = reinterpret_cast<const char *>(rk);
            uint32_t lbase[NE], run[NE] = {0, 0};
#pragma unroll
            for (int k = 0; k < NE; ++k) lbase[k] = (tid * NE + k) * 2;           // bits 0 .. 9; the level index goes into bits 10 .. 13
            const int nfull = L >> 3;
            if constexpr (LFIX == 32 || LFIX == 16) {
                constexpr int NFW = LFIX / 8;                              // full words
                uint32_t opq = 0;
                asm volatile("" : "+v"(opq));
                const uint32_t *pv = reinterpret_cast<const uint32_t *>(perm_s) + opq;
                uint32_t P[NFW][NE];
#pragma unroll
                for (int wd = 0; wd < NFW; ++wd)
#pragma unroll
                    for (int k = 0; k < NE; ++k) {
                        P[wd][k] = (cw[wd][k] + run[k]) * 0x11111111u;
                        // opaque: the compiler otherwise folds every LEFT shift of P below into a multiplication of its own
                        // (v_mul_lo_u32 by 0x440 / 0x444: quarter rate) -- 24 of them per iteration instead of 8
                        asm volatile("" : "+v"(P[wd][k]));
                        run[k] = P[wd][k] >> 28;
                    }
#pragma unroll
                for (int half = 0; half < NFW / 2; ++half) {
                    uint32_t v[16];
                    uint32_t pw[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) pw[i] = __builtin_amdgcn_readfirstlane(pv[4 * half + i]);
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int wd = 2 * half + w2;
                            uint32_t r2[NE];
#pragma unroll
                            for (int k = 0; k < NE; ++k) {
                                const uint32_t sh = j < 3 ? (P[wd][k] << (10 - 4 * j)) : (P[wd][k] >> (4 * j - 10));
                                r2[k] = *reinterpret_cast<const unsigned short *>(rkb + ((sh & 0x3c00u) | lbase[k]));
                            }
                            v[8 * w2 + j] = r2[0] | (r2[1] << 16);
                        }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const uint32_t plane = (pw[i >> 2] >> (8 * (i & 3))) & 0xffu;
                        if constexpr (BUF) {
                            // all planes within 4 GB of out_idx: one buffer descriptor, the plane's byte offset as the
                            // instruction's SCALAR offset -- two scalar instructions per store instead of eight
                            __builtin_amdgcn_raw_buffer_store_b32(v[i], orsrc, voff2, plane * plane_bytes, 2 /* nt */);
                        } else {
                            const uint16_t *pl = out_idx + (long)plane * E;          // uniform: a scalar register pair
                            asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(voff2), "v"(v[i]), "s"(pl) : "memory");
                        }
                    }
                }
            } else
#pragma unroll
            for (int wd = 0; wd < CW - 1; ++wd) {
                if (wd > nfull) break;
                // eight plane numbers: one uniform LDS read made scalar.  (Read from the kernel arguments, the 32 plane bases are
                // hoisted out of the element loop into 64 SGPRs, spilled to VGPR lanes and fetched back with two v_readlane
                // each -- a quarter of the emission's vector instructions.)
                uint32_t opq = 0;
                asm volatile("" : "+v"(opq));
                const uint32_t *pv = reinterpret_cast<const uint32_t *>(perm_s) + 2 * wd + opq;
                uint2 pw;
                pw.x = __builtin_amdgcn_readfirstlane(pv[0]);
                pw.y = __builtin_amdgcn_readfirstlane(pv[1]);
                uint32_t P[NE];
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    P[k] = (cw[wd][k] + run[k]) * 0x11111111u;
                    run[k] = P[k] >> 28;
                }
                auto addr = [&](int j, int k) {
                    const uint32_t sh = j < 3 ? (P[k] << (10 - 4 * j)) : (P[k] >> (4 * j - 10));
                    return (sh & 0x3c00u) | lbase[k];
                };
                auto plane = [&](int j) { return ((j < 4 ? pw.x : pw.y) >> (8 * (j & 3))) & 0xffu; };
                if (wd < nfull) {                                                     // a whole word: eight sweep points, no tests
                    uint32_t rank[8][NE];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int k = 0; k < NE; ++k) rank[j][k] = *reinterpret_cast<const unsigned short *>(rkb + addr(j, k));
                    // plane base in scalar registers + one 32-bit lane offset: no 64-bit address arithmetic per store
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint16_t *pl = out_idx + (long)plane(j) * E;            // uniform: lives in a scalar register pair
                        const uint32_t v = rank[j][0] | (rank[j][1] << 16);
                        asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(voff2), "v"(v), "s"(pl) : "memory");
                    }
                } else {
                    const int lim = L - wd * 8;
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        if (j >= lim) break;
                        *reinterpret_cast<uint32_t *>(oi + (long)plane(j) * E) =
                            *reinterpret_cast<const unsigned short *>(rkb + addr(j, 0)) |
                            ((uint32_t)*reinterpret_cast<const unsigned short *>(rkb + addr(j, 1)) << 16);
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
                    if (i0 + k < n_per_ch) oi[(long)sw.perm[l] * E + k] = rk[(r[k] * 256 + tid) * NE + k];
                }
            }
        }
        // ---- sweep points inside a guard band or won by a marked level: literal scan, result overwrites the emitted one
        if (__builtin_amdgcn_ballot_w64((flags[0] | flags[1]) != 0u) != 0ull) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the emitted values have landed before they are replaced
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                while (__builtin_amdgcn_ballot_w64(flags[k] != 0u) != 0ull) {
                    if (flags[k] != 0u) {
                        const int l = __builtin_ctz(flags[k]);
                        flags[k] &= flags[k] - 1u;
                        const int lo_ = perm_s[l];
                        oi[(long)lo_ * E + k] = (uint16_t)exact_rank_scan<N>(tb, m4[k], s4[k], penl + lo_ * PS);
                    }
                }
            }
        }
    }
}


// =====================================================================================================================
// K1p: the solve for ONE to FOUR lambdas per call -- the literal signature quantize(mu, sigma, lmbda) -- with exact pruning.
//
// Per bit level only the NEARER neighbour's distortion is formed (one exact quotient: the cost is monotone in |P - z| and the
// two sides of a level share their penalty, so min(cost L_n, cost R_n) = fl(min(dL, dR) + pen_n)); per lambda the running
// minimum S over the levels, the slot that attains it first and a "several levels attain it" mark are kept.  A level deeper
// than n cannot win or tie once S < min_{m > n} pen_m, because every cost is at least its penalty: the descent stops for
// the whole wave as soon as that holds for all its lanes and lambdas (at lambda = 1 after four or five of the eleven levels,
// never at lambda = 2^-8).  Afterwards the two sides of the ONE winning level are compared with the reference's rule (L
// keeps the level unless fl(dR + pen) < fl(dL + pen)), and the lanes in which several levels attained S -- the
// reference's candidate order [L_0..L_N, R_1..R_N] decides there -- take the literal scan.  No tie certificate, no
// precondition on the penalties: every comparison is one the reference makes.
constexpr int kPrunedMaxL = 2;                       // measured: 1 lambda 15-43 % faster than k_quant_fast, 2 equal, 4 slower (142 against 112 us)
template <int N, int LL>
__global__ void __launch_bounds__(256, 4)
k_quant_pruned(const float *__restrict__ mu, const float *__restrict__ sg, long n_per_ch, long ch_stride, int C,
               const float *__restrict__ table, Lambdas32 lam, const float *__restrict__ len, uint16_t *__restrict__ out_idx, long E,
               int vec_ok, int dbg) {
    constexpr int T = table_size(N);
    constexpr int N1 = N + 1;
    constexpr int NE = 2;
    __shared__ float tb[T + 1];
    __shared__ __align__(8) float2 ps[LL][N1 + 1];            // { pen[l][n], min_{m >= n} pen[l][m] }; the last row ends the suffix
    __shared__ float prow[LL][N1 + 1];                         // pen[l][n] again, contiguous: the literal scan's argument
    const int c = blockIdx.y;
    for (int i = threadIdx.x; i < T; i += blockDim.x) tb[i] = table[(long)c * T + i];
    if (threadIdx.x < LL) {
        const int l = threadIdx.x;
        float run = 3.0e38f;
        ps[l][N1] = make_float2(0.0f, run);
        for (int n = N; n >= 0; --n) {
            // pen[l][n] = fl32(lambda_l) * len[l][c][n], len = n for the raw lengths (quantizer.py:167-175, utils.py:394-396)
            const float p = __fmul_rn(lam.lam[l], len ? len[((long)l * C + c) * N1 + n] : (float)n);
            run = p < run ? p : run;                          // NaN penalties (outside the contract) never prune
            ps[l][n] = make_float2(p, run);
            prow[l][n] = p;
        }
    }
    __syncthreads();
    const bool force_slow = dbg == 1;
    const long base = (long)c * ch_stride;
    const long npairs = (n_per_ch + NE - 1) / NE;
    const char *tbb = reinterpret_cast<const char *>(tb);
    const unsigned int lane = threadIdx.x & 63u;

    unsigned int rot = wave_slot();
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < npairs; q += (long)gridDim.x * blockDim.x) {
        rotate_issue_priority(rot);                            // resident grid: see vbq_common.h
        const long i0 = q * NE;
        const bool full = (vec_ok & 1) && (i0 + NE <= n_per_ch);
        float m4[NE], s4[NE];
        if (!(vec_ok & 2) && full) {
            const float2 mv = *reinterpret_cast<const float2 *>(mu + base + i0);
            const float2 sv = *reinterpret_cast<const float2 *>(sg + base + i0);
            m4[0] = mv.x; m4[1] = mv.y;
            s4[0] = sv.x; s4[1] = sv.y;
        } else {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const bool ok = i0 + k < n_per_ch;
                const long at = (vec_ok & 2) ? (i0 + k) * C + c : base + i0 + k;       // channel-last input: see k_quant_fast
                m4[k] = ok ? mu[at] : 0.0f;
                s4[k] = ok ? sg[at] : 1.0f;
            }
        }
        uint32_t g[NE];
        double rinv[NE];
        float S[NE][LL];
        uint32_t win[NE][LL];                                  // byte slot of the visited point (bits 0 .. 12) | level << 16
        uint64_t multi[NE][LL];                                // lanes in which several levels attain S
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            g[k] = 0;
            rinv[k] = __ddiv_rn(1.0, (double)s4[k]);
#pragma unroll
            for (int l = 0; l < LL; ++l) { S[k][l] = 0.0f; win[k][l] = 0; multi[k][l] = 0; }
        }
        // The penalties are read from LDS at every level, through an index the compiler cannot see through: hoisted out of
        // the element loop (they are loop-invariant) they would occupy 22 registers per lambda and spill.
        uint32_t opq = 0;
        asm volatile("" : "+v"(opq));
        const float *psv = reinterpret_cast<const float *>(&ps[0][0]) + opq;
#pragma unroll
        for (int n = 0; n <= N; ++n) {
            const int off4 = 4 * ((1 << n) - 1);
            const int top4 = off4;
            uint64_t active = 0;
            float pen[LL], stop[LL];                           // this level's penalty; min penalty of every deeper level
#pragma unroll
            for (int l = 0; l < LL; ++l) {
                pen[l] = psv[2 * (l * (N1 + 1) + n)];
                stop[l] = psv[2 * (l * (N1 + 1) + n + 1) + 1];
            }
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const float pj = *reinterpret_cast<const float *>(tbb + off4 + g[k]);
                const bool below = pj < m4[k];
                float dmin;
                if (n == 0) {
                    dmin = __fsub_rn(pj, m4[k]);
                } else {
                    int o4 = (int)g[k] + (below ? 4 : -4);     // the other neighbour sits on z's side of the visited point
                    o4 = o4 < 0 ? 0 : (o4 > top4 ? top4 : o4);
                    const float po = *reinterpret_cast<const float *>(tbb + off4 + o4);
                    dmin = fminf(fabsf(__fsub_rn(pj, m4[k])), fabsf(__fsub_rn(po, m4[k])));
                }
                const float t = (float)__dmul_rn((double)dmin, rinv[k]);
                const float du = __fmul_rn(0.5f, __fmul_rn(t, t));
                const uint32_t here = g[k] | ((uint32_t)n << 16);
#pragma unroll
                for (int l = 0; l < LL; ++l) {
                    const float cst = __fadd_rn(du, pen[l]);
                    if (n == 0) {
                        S[k][l] = cst;
                        win[k][l] = here;
                    } else {
                        const bool lt = cst < S[k][l];
                        const uint64_t eq = __builtin_amdgcn_ballot_w64(cst == S[k][l]);
                        multi[k][l] = (multi[k][l] & ~__builtin_amdgcn_ballot_w64(lt)) | eq;
                        S[k][l] = lt ? cst : S[k][l];
                        win[k][l] = lt ? here : win[k][l];
                    }
                    if (n < N) active |= __builtin_amdgcn_ballot_w64(!(S[k][l] < stop[l]));
                }
                g[k] = 2 * g[k] + (below ? 4u : 0u);
            }
            if (n < N && active == 0 && n >= 1) break;          // no deeper level can win or tie in this wave
        }
        // the winning level's two sides, the reference's rule between them; several levels at S: the literal scan
        uint32_t rank[LL][NE];
#pragma unroll
        for (int k = 0; k < NE; ++k)
#pragma unroll
            for (int l = 0; l < LL; ++l) {
                const uint32_t n = win[k][l] >> 16;
                const uint32_t gj = win[k][l] & 0xffffu;
                const uint32_t off4 = 4u * ((1u << n) - 1u), top4 = off4;
                const float pj = *reinterpret_cast<const float *>(tbb + off4 + gj);
                const bool below = pj < m4[k];
                // interval of the level as quantizer.py:75-76 forms it: L = max(G - 1, 0), R = min(G, 2^n - 1), G = slot + [p < z];
                // on the deepest level the grid has no padding and L stays one slot back above the last point (:54-57)
                const uint32_t G4 = gj + (below ? 4u : 0u);
                uint32_t lo4 = G4 >= 4u ? G4 - 4u : 0u;
                if (n == (uint32_t)N) lo4 = lo4 > top4 - 4u ? top4 - 4u : lo4;
                const uint32_t hi4 = G4 > top4 ? top4 : G4;
                const float pen = psv[2 * (l * (N1 + 1) + n)];
                const float cL = __fadd_rn(dist_cost(*reinterpret_cast<const float *>(tbb + off4 + lo4), m4[k], rinv[k]), pen);
                const float cR = __fadd_rn(dist_cost(*reinterpret_cast<const float *>(tbb + off4 + hi4), m4[k], rinv[k]), pen);
                const uint32_t b4 = (n >= 1u && cR < cL) ? hi4 : lo4;             // level 0 has one point
                rank[l][k] = (((b4 >> 1) + 1u) << ((uint32_t)N - n)) - 1u;         // ((2 pos + 1) << (N - n)) - 1, pos = b4 / 4
            }
        uint64_t any_multi = force_slow ? ~0ull : 0ull;
#pragma unroll
        for (int k = 0; k < NE; ++k)
#pragma unroll
            for (int l = 0; l < LL; ++l) any_multi |= multi[k][l];
        if (any_multi != 0) {
#pragma unroll
            for (int k = 0; k < NE; ++k)
#pragma unroll
                for (int l = 0; l < LL; ++l)
                    if (((multi[k][l] >> lane) & 1ull) || force_slow) rank[l][k] = exact_rank_scan<N>(tb, m4[k], s4[k], prow[l]);
        }
#pragma unroll
        for (int l = 0; l < LL; ++l) {
            uint16_t *o = out_idx + (long)l * E + base + i0;
            if (full) {
                *reinterpret_cast<uint32_t *>(o) = rank[l][0] | (rank[l][1] << 16);
            } else {
#pragma unroll
                for (int k = 0; k < NE; ++k)
                    if (i0 + k < n_per_ch) o[k] = (uint16_t)rank[l][k];
            }
        }
    }
}

}  // namespace

// The grid of K1 (k_quant_fast) for one launch: workgroups per channel, and whether all of them are resident from start to end.
// A pure function of the sizes, the device's CU count and the caller's two launch arguments (no process state).
void quant_fast_grid(int64_t n_per_ch, int32_t n_ch, int wg_per_cu, int reserved, bool counting, int64_t *out_gx, bool *out_resident) {
    const int64_t nquads = (n_per_ch + kFastNE - 1) / kFastNE;
    int64_t gx = (nquads + kFastThreads - 1) / kFastThreads;
    // A RESIDENT grid: 4 workgroups per CU (110 VGPRs: four waves per SIMD; 3 with NE=4) that stay from start to end, every
    // channel's share balanced, the issue priority rotating over them (vbq_common.h) -- 335 us on Kodak-24 against 358 us
    // for 18 short-lived workgroups per channel and 394 us for the same resident grid without the rotation.
    // wg_per_cu < 4 leaves LDS and wave slots for a kernel of another stream (K2 overlapping this launch); 5 is taken as 4
    // (a fifth workgroup per CU would only wait for a slot: 444 us).
    static const bool dynamic_env = [] { const char *e = getenv("VBQ_K1_DYNAMIC"); return e && e[0] == '1'; }();   // A/B switch
    const bool explicit_wgs = wg_per_cu >= 1 && wg_per_cu <= 5;
    const int fit = kFastNE == 4 ? 3 : 4;
    // Slots the caller reserves for a kernel of another stream (`reserved_workgroups` of the entry points: the overlapped
    // all-reduce of a sharded build).
    // A resident workgroup that finds its slot taken starts only when another one has FINISHED ALL its iterations -- up to twice
    // the kernel time (EXPERIMENTS.md, "resident grids beside a collective") -- so the grid is sized to the slots that are left;
    // the grid is (workgroups per channel) x channels, so with many channels that costs whole rounds of n_ch slots: when more
    // than a tenth of the chip would be given up the launch falls back to short-lived workgroups, which lose 8 % alone but
    // only their share of the taken slots beside a collective.
    bool dynamic_grid = dynamic_env;
    if (!dynamic_grid && !explicit_wgs && reserved > 0) {
        const int64_t all = (int64_t)num_cus() * fit;
        const int64_t kept = (resident_slots(fit, reserved) / n_ch) * n_ch;
        if (kept * 10 < all * 9) dynamic_grid = true;
    }
    const int per_cu = explicit_wgs ? (wg_per_cu < fit ? wg_per_cu : fit) : (dynamic_grid ? 5 : fit);
    int64_t cap = (dynamic_grid && !explicit_wgs ? (int64_t)num_cus() * per_cu * 4 : resident_slots(per_cu, reserved)) / n_ch;
    const bool resident = !(dynamic_grid && !explicit_wgs) && (int64_t)n_ch <= (int64_t)num_cus() * per_cu;
    if (cap < 1) cap = 1;
    if (gx > cap) {
        const int64_t iters = gx;
        if (resident) {
            // all workgroups are resident: what counts is the number of iterations of the busiest one, R = ceil(iters / cap);
            // the fewest workgroups that still need only R keep the tail balanced (as K1t's launcher)
            const int64_t rounds_needed = (iters + cap - 1) / cap;
            gx = (iters + rounds_needed - 1) / rounds_needed;
        } else {
            // every workgroup walks ceil(iters / gx) chunks: pick the gx in [cap/2, cap] that wastes the
            // fewest chunk slots (e.g. 72 chunks per channel: 18 workgroups x 4, not 20 x 3.6)
            int64_t best = cap, best_pad = ((iters + cap - 1) / cap) * cap - iters;
            for (int64_t g = cap - 1; g >= (cap + 1) / 2 && best_pad > 0; --g) {
                const int64_t pad = ((iters + g - 1) / g) * g - iters;
                if (pad * best < best_pad * g) { best = g; best_pad = pad; }
            }
            gx = best;
        }
        // count mode: a 16-bit partial counter (16 words x 2 halves per (lambda, level)) takes at most 64 / 32 lanes x the waves of
        // a workgroup x the elements of a lane per iteration
        constexpr int64_t max_iters = 65000 / ((64 / 32) * (kFastThreads / 64) * kFastNE);
        if (counting && (iters + gx - 1) / gx > max_iters) gx = (iters + max_iters - 1) / max_iters;
    }
    if (gx < 1) gx = 1;
    *out_gx = gx;
    *out_resident = resident;
}

template <int N>
int launch_quant_fast(const float *mu, const float *sg, int64_t n_per_ch, int64_t ch_stride, int32_t n_ch, const float *table,
                      const Lambdas32 &lam, const float *len, int32_t L, uint16_t *out_idx, float *out_zhat,
                      float *out_bits, int64_t E, int vec_ok,
                      unsigned long long *level_counts, int wg_per_cu, int reserved, hipStream_t st) {
    int64_t gx = 1;
    bool resident = false;
    quant_fast_grid(n_per_ch, n_ch, wg_per_cu, reserved, level_counts != nullptr, &gx, &resident);
    if (resident) vec_ok |= 4;
    const dim3 grid((unsigned)gx, (unsigned)n_ch), block(kFastThreads);
    static const int dbg = [] { const char *e = getenv("VBQ_FAST_DEBUG"); return e ? atoi(e) : 0; }();
    if (level_counts)
        hipLaunchKernelGGL((k_quant_fast<N, 2>), grid, block, 0, st, mu, sg, (long)n_per_ch, (long)ch_stride, (int)n_ch, table,
                           lam, len, (int)L, out_idx, out_zhat, out_bits, (long)E, vec_ok, dbg, level_counts);
    else if (out_zhat || out_bits)
        hipLaunchKernelGGL((k_quant_fast<N, 1>), grid, block, 0, st, mu, sg, (long)n_per_ch, (long)ch_stride, (int)n_ch, table,
                           lam, len, (int)L, out_idx, out_zhat, out_bits, (long)E, vec_ok, dbg, level_counts);
    else
        hipLaunchKernelGGL((k_quant_fast<N, 0>), grid, block, 0, st, mu, sg, (long)n_per_ch, (long)ch_stride, (int)n_ch, table,
                           lam, len, (int)L, out_idx, out_zhat, out_bits, (long)E, vec_ok, dbg, level_counts);
    VBQ_CHECK_LAUNCH("quant_fast");
    return VBQ_OK;
}

// Host side of K1p: one to four lambdas, indices the only output.  Returns 1 when the call is not its kind.
template <int N>
int launch_quant_pruned(const float *mu, const float *sg, int64_t n_per_ch, int64_t ch_stride, int32_t n_ch, const float *table,
                        const Lambdas32 &lam, const float *len, int32_t L, uint16_t *out_idx, int64_t E, int vec_ok, hipStream_t st) {
    static const bool off = [] { const char *e = getenv("VBQ_NO_PRUNED"); return e && e[0] == '1'; }();
    if (off || L < 1 || L > kPrunedMaxL) return 1;
    const int64_t npairs = (n_per_ch + 1) / 2;
    int64_t gx = (npairs + 255) / 256;
    int64_t cap = (int64_t)num_cus() * 4 / n_ch;               // the 4 workgroups per CU that are resident (one round: with the
                                                                // priority rotation 71 us at lambda = 2^-8, two rounds 79, no rotation 74)
    if (cap < 1) cap = 1;
    if (gx > cap) gx = cap;
    static const int dbg = [] { const char *e = getenv("VBQ_FAST_DEBUG"); return e ? atoi(e) : 0; }();
    const dim3 grid((unsigned)gx, (unsigned)n_ch), block(256);
#define VBQ_PRUNED_CASE(LLv)                                                                                          \
    case LLv:                                                                                                          \
        hipLaunchKernelGGL((k_quant_pruned<N, LLv>), grid, block, 0, st, mu, sg, (long)n_per_ch, (long)ch_stride, (int)n_ch, \
                           table, lam, len, out_idx, (long)E, vec_ok, dbg);                                            \
        break;
    switch (L) {
        VBQ_PRUNED_CASE(1) VBQ_PRUNED_CASE(2) VBQ_PRUNED_CASE(3) VBQ_PRUNED_CASE(4)
        default: return 1;
    }
#undef VBQ_PRUNED_CASE
    VBQ_CHECK_LAUNCH("quant_pruned");
    return VBQ_OK;
}


// Sort the sweep by its f32 values and build the bucket table.  false: the sweep is not eligible for the threshold kernels
// (more than 32 values, a value outside the fast kernels' range, two values in one bucket, more than 16 octaves).
static bool build_hull_sweep(const double *lam, int L, HullSweep &sw) {
    if (L < 1 || L > 32) return false;
    int order[32];
    for (int i = 0; i < L; ++i) order[i] = i;
    for (int i = 1; i < L; ++i)                              // insertion sort by the f32 value
        for (int j = i; j > 0 && (float)lam[order[j]] < (float)lam[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    int prev_key = -1;
    for (int i = 0; i < 32; ++i) { sw.lam[i] = kHullBig; sw.perm[i] = 0; }
    for (int i = 0; i < L; ++i) {
        const float v = (float)lam[order[i]];
        if (!(v >= 1.9e-12f && v <= 1.8e19f)) return false;
        uint32_t bits;
        memcpy(&bits, &v, 4);
        const int key = (int)(bits >> 16);
        if (key <= prev_key) return false;                    // two sweep points in one bucket (or equal): dense kernel
        prev_key = key;
        sw.lam[i] = v;
        sw.perm[i] = (unsigned char)order[i];
    }
    uint32_t b0;
    memcpy(&b0, &sw.lam[0], 4);
    sw.key0 = (int)(b0 >> 16);
    sw.nkeys = prev_key - sw.key0 + 2;
    sw.L = L;
    if (sw.nkeys > kHullKeys) return false;
    int l = 0;
    for (int b = 0; b < kHullKeys; ++b) {
        while (l < L) {
            uint32_t bits;
            memcpy(&bits, &sw.lam[l], 4);
            if ((int)(bits >> 16) < sw.key0 + b) ++l; else break;
        }
        sw.lut[b] = (unsigned char)l;
    }
    return true;
}

// The grid of K1t (k_level_counts_hull) for one launch: workgroups per channel (all resident).  Pure, as quant_fast_grid.
int64_t level_counts_hull_grid(int64_t n_per_ch, int32_t n_ch, int reserved) {
    const int64_t nquads = (n_per_ch + VBQ_HULL_NE - 1) / VBQ_HULL_NE;
    int64_t gx = (nquads + kHullThreads - 1) / kHullThreads;
    // grid = the resident workgroups, one round (measured).  With slots reserved for another stream's kernel (see
    // launch_quant_fast) the grid shrinks to what is left when the (workgroups x channels) shape allows it without giving up
    // more than a tenth of the chip; otherwise it drops to three workgroups per CU: 129 us alone instead of 117, but 146-150
    // instead of 172 beside a neighbour (short-lived workgroups: 156 alone, 172-178 beside one; two per CU: 160 either way --
    // EXPERIMENTS.md, "resident grids beside a collective").
    constexpr int per_cu = VBQ_HULL_WAVES * 256 / kHullThreads;
    constexpr int rounds = 1;
    int64_t slots = resident_slots(per_cu, reserved);
    // (only with slots reserved: without a reservation the grid is floor(all / n_ch) workgroups per channel, whatever n_ch)
    if (reserved > 0 && (slots / n_ch) * n_ch * 10 < (int64_t)num_cus() * per_cu * 9)
        slots = (int64_t)num_cus() * (per_cu > 1 ? per_cu - 1 : 1);
    static const int exp_per_cu = [] { const char *e = getenv("VBQ_HULL_WG_PER_CU"); return e ? atoi(e) : 0; }();   // A/B switch
    if (exp_per_cu > 0 && exp_per_cu < per_cu) slots = (int64_t)num_cus() * exp_per_cu;
    int64_t cap = slots * rounds / n_ch;                    // VBQ_HULL_WAVES x 4 waves per CU resident
    if (cap < 1) cap = 1;
    if (gx > cap) {
        // All `cap` workgroups are resident at once and the kernel is latency-bound, so what counts is the number of
        // iterations the busiest workgroup runs, R = ceil(iters / cap); the fewest workgroups that still need only R keep
        // the tail balanced.  (Minimising the padding alone once picked 514 workgroups x 38 iterations for 1e7 elements --
        // half the wave slots empty, 221 us instead of 160.)
        const int64_t iters = gx;
        const int64_t rounds_needed = (iters + cap - 1) / cap;
        gx = (iters + rounds_needed - 1) / rounds_needed;
        // 16-bit partial counters: a half takes at most (32 / copies) lanes x 4 waves x NE per iteration
        const int64_t max_iters = 65000 / ((32 / VBQ_HULL_COPIES) * (kHullThreads / 64) * VBQ_HULL_NE);
        if ((iters + gx - 1) / gx > max_iters) gx = (iters + max_iters - 1) / max_iters;
    }
    return gx < 1 ? 1 : gx;
}

// Host side of K1t: sort the sweep, check that the bucket table applies (distinct f32 values at least one bucket
// apart, within 16 octaves), launch.  Returns 1 when the sweep is not eligible (the caller then takes the dense
// counting kernel), VBQ_OK / an error otherwise.
int launch_level_counts_hull10(const float *mu, const float *sg, int64_t n_per_ch, int64_t ch_stride, int32_t n_ch,
                               const float *table, const double *lam, int32_t L, int vec_ok,
                               unsigned long long *level_counts, int reserved, hipStream_t st) {
    static const bool off = [] { const char *e = getenv("VBQ_NO_HULL"); return e && e[0] == '1'; }();
    if (off || L < 1 || L > 32) return 1;
    HullSweep sw;
    if (!build_hull_sweep(lam, L, sw)) return 1;
    const int64_t gx = level_counts_hull_grid(n_per_ch, n_ch, reserved);
    static const int dbg = [] { const char *e = getenv("VBQ_FAST_DEBUG"); return e ? atoi(e) : 0; }();
    Lambdas32 l32;
    for (int i = 0; i < kMaxLambdaChunk; ++i) l32.lam[i] = i < L ? (float)lam[i] : 0.0f;
    hipLaunchKernelGGL((k_level_counts_hull<10>), dim3((unsigned)gx, (unsigned)n_ch), dim3(kHullThreads), 0, st, mu, sg,
                       (long)n_per_ch, (long)ch_stride, (int)n_ch, table, l32, sw, vec_ok, level_counts, dbg);
    VBQ_CHECK_LAUNCH("level_counts_hull");
    return VBQ_OK;
}

// Host side of K1e.  Returns 1 when the sweep is not eligible (the caller takes k_quant_fast).
int launch_quant_hull_idx10(const float *mu, const float *sg, int64_t n_per_ch, int64_t ch_stride, int32_t n_ch,
                            const float *table, const double *lam, int32_t L, int vec_ok, uint16_t *out_idx, int64_t E,
                            hipStream_t st) {
    static const bool off = [] { const char *e = getenv("VBQ_NO_HULL"); return e && e[0] == '1'; }();
    if (off || L < 16) return 1;                             // below that the thresholds cost more than the solves they replace
                                                             // (Kodak-24: 0.25 against 0.27 ms at 16 lambdas, 0.29 against 0.43 at 32)
    // The emission addresses an index plane with a 32-bit BYTE offset from a scalar plane base (global_store_dword voff, v,
    // s[plane]): planes of 2^31 elements or more (a 16-lambda sweep of that size is 64 GB of indices: it fits) go to K1.
    if (2 * (uint64_t)E > 0xffffffffull) return 1;
    HullSweep sw;
    if (!build_hull_sweep(lam, L, sw)) return 1;
    const int64_t npairs = (n_per_ch + 1) / 2;
    int64_t gx = (npairs + 255) / 256;
    constexpr int rounds = 2;                               // grid = this many times the resident workgroups (measured)
    int64_t cap = (int64_t)num_cus() * VBQ_K1E_WAVES * rounds / n_ch;          // VBQ_K1E_WAVES workgroups per CU resident
    if (cap < 1) cap = 1;
    if (gx > cap) gx = cap;
    static const int dbg = [] { const char *e = getenv("VBQ_FAST_DEBUG"); return e ? atoi(e) : 0; }();
    Lambdas32 l32;
    for (int i = 0; i < kMaxLambdaChunk; ++i) l32.lam[i] = i < L ? (float)lam[i] : 0.0f;
    // all L planes within 4 GB: the emission addresses them through one buffer descriptor (32-bit scalar plane offsets)
    const bool buf = (uint64_t)L * 2ull * (uint64_t)E <= 0xffffffffull;
#define VBQ_K1E_LAUNCH(LF, BF)                                                                                                \
    hipLaunchKernelGGL((k_quant_hull_idx<10, LF, BF>), dim3((unsigned)gx, (unsigned)n_ch), dim3(256), 0, st, mu, sg,          \
                       (long)n_per_ch, (long)ch_stride, (int)n_ch, table, l32, sw, vec_ok, out_idx, (long)E, dbg)
    if (L == 32) { if (buf) VBQ_K1E_LAUNCH(32, true); else VBQ_K1E_LAUNCH(32, false); }
    else if (L == 16) { if (buf) VBQ_K1E_LAUNCH(16, true); else VBQ_K1E_LAUNCH(16, false); }      // the sweep of post_process.py:115
    else VBQ_K1E_LAUNCH(0, false);
#undef VBQ_K1E_LAUNCH
    VBQ_CHECK_LAUNCH("quant_hull_idx");
    return VBQ_OK;
}

#define VBQ_INST_FAST(NN)                                                                                              \
    template int launch_quant_fast<NN>(const float *, const float *, int64_t, int64_t, int32_t, const float *,        \
                                       const Lambdas32 &, const float *, int32_t, uint16_t *, float *, float *, int64_t, \
                                       int, unsigned long long *, int, int, hipStream_t);
VBQ_FOR_EACH_N(VBQ_INST_FAST)
#undef VBQ_INST_FAST
#define VBQ_INST_PRUNED(NN)                                                                                           \
    template int launch_quant_pruned<NN>(const float *, const float *, int64_t, int64_t, int32_t, const float *,     \
                                         const Lambdas32 &, const float *, int32_t, uint16_t *, int64_t, int, hipStream_t);
VBQ_FOR_EACH_N(VBQ_INST_PRUNED)
#undef VBQ_INST_PRUNED

}  // namespace vbq

// The launch shape the two resident solve kernels would take: a query, nothing is launched (tests / harness output).
extern "C" int vbq_solve_grid(int32_t kernel, int64_t n_rows, int32_t n_ch, int32_t workgroups_per_cu, int32_t reserved_workgroups,
                              int64_t *h_grid) {
    using namespace vbq;
    VBQ_REQUIRE(h_grid && n_rows >= 1 && n_ch >= 1 && n_ch <= 65535, VBQ_ERR_INVALID_ARGUMENT, "vbq_solve_grid: bad sizes or null pointer");
    VBQ_REQUIRE(kernel == VBQ_GRID_K1 || kernel == VBQ_GRID_K1T, VBQ_ERR_INVALID_ARGUMENT, "vbq_solve_grid: unknown kernel %d", kernel);
    VBQ_REQUIRE(workgroups_per_cu >= 0 && workgroups_per_cu <= 5, VBQ_ERR_INVALID_ARGUMENT, "vbq_solve_grid: workgroups_per_cu=%d not in 0..5",
                workgroups_per_cu);
    const int reserved = reserved_workgroups < 0 ? default_reserved_workgroups() : reserved_workgroups;
    if (kernel == VBQ_GRID_K1) {
        int64_t gx = 1;
        bool resident = false;
        quant_fast_grid(n_rows, n_ch, workgroups_per_cu, reserved, false, &gx, &resident);
        h_grid[0] = gx;
        h_grid[2] = resident ? 1 : 0;
    } else {
        h_grid[0] = level_counts_hull_grid(n_rows, n_ch, reserved);
        h_grid[2] = 1;
    }
    h_grid[1] = n_ch;
    return VBQ_OK;
}

// f4: the analogy evaluator of the word-embedding notebook, prediction_ranks
// (compress-trained-word-embeddings.ipynb cell 14, ipynb:199-209), as one fused MFMA GEMM:
//
//     normed   = emb / (1e-8 + sqrt(sum(emb**2, axis=1)))                     [V, K]
//     pred_i   = normed[b_i] - normed[a_i] + normed[c_i]                       [Nq, K]
//     score_iv = pred_i . normed_v                                             [Nq, V]  (never materialised)
//     rank_i   = V - #{v : score_iv < score_i,d_i} - 1
//
// The Nq x V score matrix (2e9 entries for the notebook's 19.5k questions x 100k words) is
// produced tile by tile on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32, one rounding per
// product, accumulated in k order) and consumed in registers by the comparison with the row's
// ground-truth score; only Nq counters leave the chip.  gfx950 / CDNA4 only.
//
// Numerics: an f32 MFMA accumulator is bit for bit the chain fma(a_k, b_k, acc) over ascending k
// (measured on gfx950), so the ground-truth score is computed by the same chain on the vector
// ALU (k_pred_gt) and is bit-identical to the tile entry it is compared with: a word never counts
// against itself, exactly as in the reference where gt is read back from the product.
#include "vbq_common.h"

namespace vbq {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));      // native vector (HIP's float4 struct kept the staging arrays in scratch)

constexpr int kBM = 128;          // questions per workgroup tile
constexpr int kBN = 128;          // words per workgroup tile
#ifndef VBQ_RANK_BK
#define VBQ_RANK_BK 32
#endif
#ifndef VBQ_RANK_WGS
#define VBQ_RANK_WGS 2
#endif
constexpr int kBK = VBQ_RANK_BK;  // k per LDS stage (32: 2 stages x 2 operands x 16 KB = 64 KB of LDS, 2 workgroups per CU)
constexpr int kRankThreads = 256; // 4 waves, each a 64 x 64 quadrant = 2 x 2 MFMA tiles

// normed^T [Kp][Vp] and the denominators 1e-8 + |emb_v| [Vp].  Row norms are f32 sums in index order
// (NumPy's pairwise sum differs in the last bits; the test tolerance covers it).  A workgroup owns 64 words:
// the embedding is read in coalesced 64 x 64 slabs through LDS, thread r < 64 adds up row r in k order,
// and the normalised slab leaves transposed, again coalesced.  Zero padding for k >= K and v >= V.
__global__ void __launch_bounds__(256)
k_normalize_t(const float *__restrict__ emb, long V, int K, int Kp, long Vp, float *__restrict__ nT, float *__restrict__ den) {
    __shared__ float tile[64][65];
    __shared__ float dn[64];
    const long v0 = (long)blockIdx.x * 64;
    const int tid = threadIdx.x;
    float s = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 64) {
        for (int i = tid; i < 64 * 64; i += 256) {
            const int r = i >> 6, j = i & 63;
            tile[r][j] = (v0 + r < V && k0 + j < K) ? emb[(v0 + r) * K + k0 + j] : 0.0f;
        }
        __syncthreads();
        if (tid < 64) {
            const int kn = K - k0 < 64 ? K - k0 : 64;
            for (int j = 0; j < kn; ++j) s = __fadd_rn(s, __fmul_rn(tile[tid][j], tile[tid][j]));
        }
        __syncthreads();
    }
    if (tid < 64) {
        dn[tid] = __fadd_rn(1e-8f, sqrtf(s));
        if (v0 + tid < Vp) den[v0 + tid] = dn[tid];
    }
    __syncthreads();
    for (int k0 = 0; k0 < Kp; k0 += 64) {
        for (int i = tid; i < 64 * 64; i += 256) {
            const int r = i >> 6, j = i & 63;
            tile[r][j] = (v0 + r < V && k0 + j < K) ? __fdiv_rn(emb[(v0 + r) * K + k0 + j], dn[r]) : 0.0f;
        }
        __syncthreads();
        for (int i = tid; i < 64 * 64; i += 256) {
            const int j = i >> 6, r = i & 63;
            if (k0 + j < Kp && v0 + r < Vp) nT[(long)(k0 + j) * Vp + v0 + r] = tile[r][j];
        }
        __syncthreads();
    }
}

// pred^T [Kp][Qp] and the ground-truth score of every question (same fma chain as the MFMA).  The four
// words of a question are re-normalised from their embedding rows (contiguous) with the stored denominators:
// the same IEEE division as in k_normalize_t, hence the same numbers as in normed^T.
__global__ void __launch_bounds__(256)
k_pred_gt(const float *__restrict__ emb, const float *__restrict__ den, int K, int Kp, const int32_t *__restrict__ an, long Q,
          long Qp, float *__restrict__ pT, float *__restrict__ gt) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Qp) return;
    if (i >= Q) {
        for (int k = 0; k < Kp; ++k) pT[(long)k * Qp + i] = 0.0f;
        gt[i] = 0.0f;
        return;
    }
    const long a = an[4 * i], b = an[4 * i + 1], c = an[4 * i + 2], d = an[4 * i + 3];
    const float *ea = emb + a * K, *eb = emb + b * K, *ec = emb + c * K, *ed = emb + d * K;
    const float da = den[a], db = den[b], dc = den[c], dd = den[d];
    float acc = 0.0f;
    for (int k = 0; k < Kp; ++k) {
        float p = 0.0f, t = 0.0f;
        if (k < K) {
            p = __fadd_rn(__fsub_rn(__fdiv_rn(eb[k], db), __fdiv_rn(ea[k], da)), __fdiv_rn(ec[k], dc));   // (nb - na) + nc
            t = __fdiv_rn(ed[k], dd);
        }
        pT[(long)k * Qp + i] = p;
        acc = __fmaf_rn(p, t, acc);
    }
    gt[i] = acc;
}

// One workgroup: 128 questions x a range of 128-word tiles.  LDS holds [k][m] slabs so that a
// wave's operand fetch (lane l: row l & 31, k = l >> 5) is one conflict-free ds_read_b32.
__global__ void __launch_bounds__(kRankThreads, VBQ_RANK_WGS)
k_rank_gemm(const float *__restrict__ pT, const float *__restrict__ nT, const float *__restrict__ gt, long Qp, long Vp,
            long V, int Kp, int K2, int tiles_per_wg, int *__restrict__ below) {
    __shared__ __align__(16) float As[2][kBK][kBM];
    __shared__ __align__(16) float Bs[2][kBK][kBN];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;          // the wave's quadrant
    const int lr = lane & 31, lh = lane >> 5;
    const long q0 = (long)blockIdx.y * kBM;
    const long nt = Vp / kBN;
    const long t0 = (long)blockIdx.x * tiles_per_wg;
    const long t1 = t0 + tiles_per_wg < nt ? t0 + tiles_per_wg : nt;

    // ground-truth scores of the workgroup's 128 questions (read back in the epilogue of every tile)
    __shared__ float gts[kBM];
    if (tid < kBM) gts[tid] = gt[q0 + tid];
    int cnt[2][16];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int i = 0; i < 16; ++i) cnt[rt][i] = 0;

    // global -> LDS: a stage is kBK rows of 128 floats per operand; thread -> rows ldr + 8 h, one float4 each
    constexpr int LH = kBK / 8;
    const int ldr = tid >> 5, ldc = (tid & 31) * 4;
    const int nk = Kp / kBK;
    const long steps = (t1 - t0) * nk;                               // flat (tile, k-chunk) sequence

    f32x4 ra[LH], rb[LH];
#pragma unroll
    for (int h = 0; h < LH; ++h) {
        ra[h] = *reinterpret_cast<const f32x4 *>(pT + (long)(ldr + 8 * h) * Qp + q0 + ldc);
        rb[h] = *reinterpret_cast<const f32x4 *>(nT + (long)(ldr + 8 * h) * Vp + t0 * kBN + ldc);
    }
    f32x16 acc[2][2];
    long t = t0;
    int kc = 0;
    for (long s = 0; s < steps; ++s) {
        const int buf = (int)(s & 1);
        if (kc == 0) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[rt][ct][i] = 0.0f;
        }
#pragma unroll
        for (int h = 0; h < LH; ++h) {
            *reinterpret_cast<f32x4 *>(&As[buf][ldr + 8 * h][ldc]) = ra[h];
            *reinterpret_cast<f32x4 *>(&Bs[buf][ldr + 8 * h][ldc]) = rb[h];
        }
        __syncthreads();                 // stage s visible; every wave is past its reads of stage s - 1 (other buffer)
        // the next stage (possibly the first of the next tile) is fetched while this one is multiplied
        const bool last_k = kc + 1 == nk;
        const long tn = last_k ? t + 1 : t;
        const int kn = last_k ? 0 : kc + 1;
        if (s + 1 < steps) {
#pragma unroll
            for (int h = 0; h < LH; ++h) {
                const long kr = (long)kn * kBK + ldr + 8 * h;
                ra[h] = *reinterpret_cast<const f32x4 *>(pT + kr * Qp + q0 + ldc);
                rb[h] = *reinterpret_cast<const f32x4 *>(nT + kr * Vp + tn * kBN + ldc);
            }
        }
        // operand fetch for step kk + 2 is issued before the four MFMAs of step kk (64 cycles each);
        // the zero padding of the last stage (k >= K rounded up to even) is skipped, not multiplied
        const int kend = last_k ? K2 - kc * kBK : kBK;               // wave-uniform
        float a[2], b[2], an[2], bn[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) a[rt] = As[buf][lh][wr + rt * 32 + lr];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) b[ct] = Bs[buf][lh][wc + ct * 32 + lr];
#pragma unroll
        for (int kk = 0; kk < kBK; kk += 2) {
            if (kk < kend) {
                if (kk + 2 < kBK) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) an[rt] = As[buf][kk + 2 + lh][wr + rt * 32 + lr];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) bn[ct] = Bs[buf][kk + 2 + lh][wc + ct * 32 + lr];
                }
                __builtin_amdgcn_sched_barrier(0);                  // keep the fetch ahead of the MFMAs
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt], b[ct], acc[rt][ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) { a[rt] = an[rt]; b[rt] = bn[rt]; }
            }
        }
        if (last_k) {
            // epilogue: count the words that score strictly below the row's ground truth
            const long v0 = t * kBN;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const bool col_ok = v0 + wc + ct * 32 + lr < V;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        cnt[rt][i] += (col_ok && acc[rt][ct][i] < gts[wr + rt * 32 + 8 * (i >> 2) + 4 * lh + (i & 3)]) ? 1 : 0;
            }
        }
        t = tn;
        kc = kn;
    }

    // sum over the 32 lanes that hold one row (same lane >> 5), then one atomic per row
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int c = cnt[rt][i];
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) c += __shfl_xor(c, m, 64);
            if (lr == 0 && c) atomicAdd(&below[q0 + wr + rt * 32 + 8 * (i >> 2) + 4 * lh + (i & 3)], c);
        }
}

__global__ void k_rank_finish(const int *__restrict__ below, long Q, long V, int64_t *__restrict__ ranks) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Q) ranks[i] = V - (int64_t)below[i] - 1;                // ipynb cell 14, last line
}

struct RankLayout {
    long Vp, Qp;
    int Kp;
    size_t off_nT, off_pT, off_gt, off_below, off_den, total;
};

RankLayout rank_layout(int64_t V, int32_t K, int64_t Q) {
    RankLayout r;
    r.Vp = (V + kBN - 1) / kBN * kBN;
    r.Qp = (Q + kBM - 1) / kBM * kBM;
    r.Kp = (K + kBK - 1) / kBK * kBK;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    r.off_nT = 0;
    r.off_pT = al(r.off_nT + sizeof(float) * (size_t)r.Kp * r.Vp);
    r.off_gt = al(r.off_pT + sizeof(float) * (size_t)r.Kp * r.Qp);
    r.off_below = al(r.off_gt + sizeof(float) * (size_t)r.Qp);
    r.off_den = al(r.off_below + sizeof(int) * (size_t)r.Qp);
    r.total = al(r.off_den + sizeof(float) * (size_t)r.Vp);
    return r;
}

}  // namespace
}  // namespace vbq

extern "C" size_t vbq_analogy_ranks_workspace_bytes(int64_t V, int32_t K, int64_t Q) {
    if (V <= 0 || K <= 0 || Q < 0) return 0;
    return vbq::rank_layout(V, K, Q).total;
}

extern "C" int vbq_analogy_ranks_f32(const float *d_emb, int64_t V, int32_t K, const int32_t *d_analogies, int64_t Q,
                                     int64_t *d_out_ranks, void *d_workspace, size_t workspace_bytes, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(V > 0 && V < (1ll << 31) && K > 0 && K <= 65536 && Q >= 0, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_analogy_ranks_f32: bad shape V=%lld K=%d Q=%lld", (long long)V, K, (long long)Q);
    if (Q == 0) return VBQ_OK;
    VBQ_REQUIRE(d_emb && d_analogies && d_out_ranks && d_workspace, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_analogy_ranks_f32: null pointer");
    const RankLayout r = rank_layout(V, K, Q);
    VBQ_REQUIRE(workspace_bytes >= r.total, VBQ_ERR_WORKSPACE, "vbq_analogy_ranks_f32: workspace %zu < %zu bytes",
                workspace_bytes, r.total);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *ws = reinterpret_cast<char *>(d_workspace);
    float *nT = reinterpret_cast<float *>(ws + r.off_nT), *pT = reinterpret_cast<float *>(ws + r.off_pT);
    float *gt = reinterpret_cast<float *>(ws + r.off_gt);
    int *below = reinterpret_cast<int *>(ws + r.off_below);
    float *den = reinterpret_cast<float *>(ws + r.off_den);
    if (hipMemsetAsync(below, 0, sizeof(int) * (size_t)r.Qp, st) != hipSuccess) {
        set_error("vbq_analogy_ranks_f32: hipMemsetAsync failed");
        return VBQ_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(k_normalize_t, dim3((unsigned)(r.Vp / 64)), dim3(256), 0, st, d_emb, (long)V, (int)K, r.Kp, r.Vp, nT, den);
    hipLaunchKernelGGL(k_pred_gt, dim3((unsigned)((r.Qp + 255) / 256)), dim3(256), 0, st, d_emb, den, (int)K, r.Kp, d_analogies,
                       (long)Q, r.Qp, pT, gt);
    // question blocks x word-tile ranges.  2 workgroups fit a CU; pick the split whose last wave of
    // workgroups wastes the least: time ~ ceil(qb * splits / 512) * tiles_per_wg
    const long qb = r.Qp / kBM, nt = r.Vp / kBN;
    long best_s = 1, best_cost = -1;
    const long slots = (long)num_cus() * VBQ_RANK_WGS;           // workgroups resident at once
    for (long s = 1; s <= nt && s <= 256; ++s) {
        const long tpw = (nt + s - 1) / s;
        const long cost = ((qb * s + slots - 1) / slots) * tpw;
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && tpw >= 8 && s > best_s)) { best_cost = cost; best_s = s; }
    }
    const int tiles_per_wg = (int)((nt + best_s - 1) / best_s);
    const long splits = (nt + tiles_per_wg - 1) / tiles_per_wg;
    hipLaunchKernelGGL(k_rank_gemm, dim3((unsigned)splits, (unsigned)qb), dim3(kRankThreads), 0, st, pT, nT, gt, r.Qp, r.Vp,
                       (long)V, r.Kp, (int)((K + 1) & ~1), tiles_per_wg, below);
    hipLaunchKernelGGL(k_rank_finish, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, st, below, (long)Q, (long)V, d_out_ranks);
    VBQ_CHECK_LAUNCH("analogy_ranks");
    return VBQ_OK;
}

// f2: a static-model rANS coder for the rank indices, so that the rates the
// reference only ESTIMATES as sum(-log2 freq) (quantizer.py:144,226-228) become real bits.
//
// Format (ours; the reference has no coder): every (lambda, channel) stream of n indices is cut
// into segments of `seg` symbols; each segment is an independent rANS stream (32-bit state,
// 16-bit renormalisation, PB = 15 probability bits, frequencies >= 1 summing to 2^15):
//     words[0..k-3] = renormalisation words in emission order, words[k-2], words[k-1] = final state
//     (low, high half), k = size of the segment in 16-bit words (<= seg + 2).
// Symbols are encoded last-to-first, so the decoder -- starting from the state at the END of the
// segment and consuming words backwards -- reproduces them first-to-last.  One thread per
// segment; the 64 segments of a workgroup belong to one stream and share its frequency /
// cumulative tables in LDS.  oracle/vbq_oracle.c (rans_* functions) is the bit-exact checker.
#include "vbq_common.h"

namespace vbq {
namespace {

constexpr int kPB = 15;
constexpr unsigned kRansL = 1u << 16;
constexpr int kRansThreads = 64;

// Frequencies and exclusive cumulative frequencies of one stream into LDS, as ONE u32 table fc[sym] = f | c << 16 (f >= 1,
// c < 2^15: one LDS read per symbol in the encoder) and -- for the decoder's slot search -- c alone with c[T] = 2^15.
__device__ __forceinline__ void stage_tables(const uint16_t *__restrict__ freq, int T, uint32_t *fc_l, uint16_t *c_l) {
    // exclusive prefix sum of <= 2048 frequencies by one wave: each lane owns a contiguous chunk
    const int lane = threadIdx.x;
    const int per = (T + kRansThreads - 1) / kRansThreads;
    unsigned sum = 0;
    for (int i = lane * per; i < min(T, (lane + 1) * per); ++i) sum += freq[i];
    unsigned incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    unsigned run = incl - sum;
    for (int i = lane * per; i < min(T, (lane + 1) * per); ++i) {
        const unsigned f = freq[i];
        fc_l[i] = f | (run << 16);                               // (an invalid table may overflow the 16 bits: the decoder rejects
        if (c_l) c_l[i] = (uint16_t)run;                         //  it through c_l[T] below before using any entry)
        run += f;
    }
    if (lane == 63 && c_l) c_l[T] = (uint16_t)(incl == (1u << kPB) ? incl : 0u);   // 2^15 for a valid table, 0 marks an invalid one
    __syncthreads();
}

// x / f and x % f for 1 <= f < 2^15 and x < f 2^17 (the encoder's invariant after renormalisation) without the ~35-instruction
// expansion of a 32-bit division: the quotient is below 2^17, a float estimate of it is off by at most one, and the remainder says
// which way (exact by construction: the result is verified, not trusted).
__device__ __forceinline__ void divmod_small(unsigned x, unsigned f, unsigned &q, unsigned &r) {
    q = (unsigned)(__uint2float_rn(x) * __builtin_amdgcn_rcpf(__uint2float_rn(f)));
    int rr = (int)(x - q * f);
    if (rr < 0) { rr += (int)f; --q; }
    if (rr >= (int)f) { rr -= (int)f; ++q; }
    r = (unsigned)rr;
}

// One thread per segment; a lane walks its segment from the last symbol to the first.  Symbols come in 16-byte groups of eight
// where the layout allows it (segment length and stream length multiples of eight: one load per eight symbols instead of eight
// dependent 2-byte loads from a line other lanes do not share).
__global__ void __launch_bounds__(kRansThreads)
k_rans_encode(const uint16_t *__restrict__ idx, long n, int T, int seg, int nseg, const uint16_t *__restrict__ freq,
              uint16_t *__restrict__ words, uint32_t *__restrict__ sizes) {
    __shared__ uint32_t fc_l[2048];
    const long s = blockIdx.y;                                   // stream
    stage_tables(freq + s * T, T, fc_l, nullptr);
    const int g = blockIdx.x * kRansThreads + threadIdx.x;       // segment within the stream
    if (g >= nseg) return;
    const long a = (long)g * seg;
    const long b = a + seg < n ? a + seg : n;
    const uint16_t *src = idx + s * n;
    uint16_t *out = words + (s * nseg + g) * (long)(seg + 2);
    unsigned x = kRansL;
    int k = 0;
    auto put = [&](unsigned sym) {
        const unsigned fc = fc_l[sym < (unsigned)T ? sym : 0u];  // (an index outside the table: memory-safe; vbq_index_max_u16 tells beforehand)
        const unsigned f = fc & 0xffffu, c = fc >> 16;
        if (x >= (f << (32 - kPB))) { out[k++] = (uint16_t)(x & 0xffffu); x >>= 16; }
        unsigned q, r;
        divmod_small(x, f, q, r);
        x = (q << kPB) + r + c;
    };
    long i = b;
    const bool vec = ((seg | n) % 8 == 0) && (reinterpret_cast<uintptr_t>(src) % 16 == 0);
    if (vec) {
        for (; i - 8 >= a; i -= 8) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + i - 8);
            put(v.w >> 16); put(v.w & 0xffffu); put(v.z >> 16); put(v.z & 0xffffu);
            put(v.y >> 16); put(v.y & 0xffffu); put(v.x >> 16); put(v.x & 0xffffu);
        }
    }
    for (--i; i >= a; --i) put(src[i]);
    out[k++] = (uint16_t)(x & 0xffffu);
    out[k++] = (uint16_t)(x >> 16);
    sizes[s * nseg + g] = (uint32_t)k;
}

// Untrusted input: the segment sizes and words may come from a damaged or foreign file.  Every read is kept
// inside the segment's (seg + 2)-word buffer and every decoded symbol below T; what is wrong is reported in
// *status (bit 0: a segment size outside [2, seg + 2]; bit 1: a segment ran out of words; bit 2: words left
// over or a final state other than the encoder's start state; bit 3: a frequency row that does not sum to 2^15).
__global__ void __launch_bounds__(kRansThreads)
k_rans_decode(const uint16_t *__restrict__ words, const uint32_t *__restrict__ sizes, long n, int T, int seg, int nseg,
              const uint16_t *__restrict__ freq, uint16_t *__restrict__ idx, uint32_t *__restrict__ status) {
    __shared__ uint32_t fc_l[2048];
    __shared__ uint16_t c_l[2049 + 1];
    // start[b] = the symbol whose slot range contains slot 16 b: the search for a slot begins there and walks
    // up (a bucket of 16 slots holds 1 symbol on average), instead of 11 dependent LDS reads of a bisection
    __shared__ uint16_t start[(1 << kPB) / 16];
    const long s = blockIdx.y;
    stage_tables(freq + s * T, T, fc_l, c_l);
    const bool table_ok = c_l[T] == (uint16_t)(1u << kPB);
    if (table_ok) {
        for (int bkt = threadIdx.x; bkt < (1 << kPB) / 16; bkt += kRansThreads) {
            const unsigned slot = 16u * bkt;
            int lo = 0, hi = T;                                  // last symbol with cum <= slot
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (c_l[mid] <= slot) lo = mid; else hi = mid;
            }
            start[bkt] = (uint16_t)lo;
        }
    }
    __syncthreads();
    const int g = blockIdx.x * kRansThreads + threadIdx.x;
    if (g >= nseg) return;
    const long a = (long)g * seg;
    const long b = a + seg < n ? a + seg : n;
    const uint16_t *in = words + (s * nseg + g) * (long)(seg + 2);
    uint16_t *dst = idx + s * n;
    unsigned bad = table_ok ? 0u : 8u;
    const unsigned k0 = sizes[s * nseg + g];
    if (k0 < 2u || k0 > (unsigned)seg + 2u) bad |= 1u;
    if (bad) {
        for (long i = a; i < b; ++i) dst[i] = 0;
        if (status) atomicOr(status, bad);
        return;
    }
    int k = (int)k0;
    unsigned x = ((unsigned)in[k - 1] << 16) | in[k - 2];
    k -= 2;
    bool starved = false;
    auto get = [&]() -> unsigned {                               // one symbol; after a starved stream: zeros
        if (starved) return 0u;
        const unsigned slot = x & ((1u << kPB) - 1u);
        unsigned lo = start[slot >> 4];                          // last symbol with cum <= slot
        while (c_l[lo + 1] <= slot) ++lo;                        // c_l[T] = 2^15 > slot ends the walk below T
        const unsigned fc = fc_l[lo];
        x = (fc & 0xffffu) * (x >> kPB) + slot - (fc >> 16);
        if (x < kRansL) {
            if (k == 0) { bad |= 2u; starved = true; }           // a valid stream never renormalises past its first word
            else x = (x << 16) | in[--k];
        }
        return lo;
    };
    long i = a;
    // eight symbols per 16-byte store where the layout allows it (instead of eight 2-byte stores into a line of its own)
    if (((seg | n) % 8 == 0) && (reinterpret_cast<uintptr_t>(dst) % 16 == 0)) {
        for (; i + 8 <= b; i += 8) {
            uint4 v;
            v.x = get(); v.x |= get() << 16;
            v.y = get(); v.y |= get() << 16;
            v.z = get(); v.z |= get() << 16;
            v.w = get(); v.w |= get() << 16;
            *reinterpret_cast<uint4 *>(dst + i) = v;
        }
    }
    for (; i < b; ++i) dst[i] = (uint16_t)get();
    if (!bad && (k != 0 || x != kRansL)) bad |= 4u;              // the encoder started from kRansL with no words written
    if (bad && status) atomicOr(status, bad);
}

}  // namespace
}  // namespace vbq

extern "C" int vbq_rans_encode_u16(const uint16_t *d_idx, int64_t n_streams, int64_t n, int32_t N, int32_t seg,
                                   const uint16_t *d_freq, uint16_t *d_words, uint32_t *d_sizes, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_streams >= 0 && n >= 0 && N >= 1 && N <= 10 && seg >= 1 && seg <= 65533 && n_streams <= 65535,
                VBQ_ERR_INVALID_ARGUMENT, "vbq_rans_encode_u16: bad sizes n_streams=%lld n=%lld N=%d seg=%d",
                (long long)n_streams, (long long)n, N, seg);
    if (n_streams == 0 || n == 0) return VBQ_OK;
    VBQ_REQUIRE(d_idx && d_freq && d_words && d_sizes, VBQ_ERR_INVALID_ARGUMENT, "vbq_rans_encode_u16: null pointer argument");
    const int64_t nseg = (n + seg - 1) / seg;
    hipLaunchKernelGGL(k_rans_encode, dim3((unsigned)((nseg + kRansThreads - 1) / kRansThreads), (unsigned)n_streams),
                       dim3(kRansThreads), 0, reinterpret_cast<hipStream_t>(stream), d_idx, (long)n, table_size(N), (int)seg,
                       (int)nseg, d_freq, d_words, d_sizes);
    VBQ_CHECK_LAUNCH("rans_encode");
    return VBQ_OK;
}

extern "C" int vbq_rans_decode_u16(const uint16_t *d_words, const uint32_t *d_sizes, int64_t n_streams, int64_t n,
                                   int32_t N, int32_t seg, const uint16_t *d_freq, uint16_t *d_idx, uint32_t *d_status,
                                   void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(n_streams >= 0 && n >= 0 && N >= 1 && N <= 10 && seg >= 1 && seg <= 65533 && n_streams <= 65535,
                VBQ_ERR_INVALID_ARGUMENT, "vbq_rans_decode_u16: bad sizes n_streams=%lld n=%lld N=%d seg=%d",
                (long long)n_streams, (long long)n, N, seg);
    if (n_streams == 0 || n == 0) return VBQ_OK;
    VBQ_REQUIRE(d_idx && d_freq && d_words && d_sizes, VBQ_ERR_INVALID_ARGUMENT, "vbq_rans_decode_u16: null pointer argument");
    const int64_t nseg = (n + seg - 1) / seg;
    hipLaunchKernelGGL(k_rans_decode, dim3((unsigned)((nseg + kRansThreads - 1) / kRansThreads), (unsigned)n_streams),
                       dim3(kRansThreads), 0, reinterpret_cast<hipStream_t>(stream), d_words, d_sizes, (long)n, table_size(N),
                       (int)seg, (int)nseg, d_freq, d_idx, d_status);
    VBQ_CHECK_LAUNCH("rans_decode");
    return VBQ_OK;
}

// Host-side piece of the entropy-model build: counts -> code lengths with the reference's float32 operations
// (img-compression/quantizer.py:105-110, 141-146), as plain C that can run on the HIP runtime's callback thread.
//
// When the -log2 step cannot be tabulated (2^24 samples per histogram row or more: the large C = 1 configurations), the small
// count tables go through a stream-ordered host stage between the device stages (vbq_amd/pipeline.py).  Round 4 ran NumPy there --
// Python on the runtime's thread, which needs the GIL: a thread that holds the GIL while it blocks on the device with a host node
// pending would deadlock.  The arithmetic is four float32 operations per bin, all IEEE except one: NumPy's float32 log2 is not
// correctly rounded (it is SVML / a SIMD polynomial, not libm), and the reference's numbers are NumPy's.  So the log2 is taken by
// NumPy's OWN inner loop, called through the function pointer the ufunc object holds (numpy/ufuncobject.h: PyUFuncGenericFunction;
// NumPy itself runs these loops without the GIL), and everything else is restated here:
//     c     = float32(count) + float32(add_n)                                   np.array(counts, float32); c += add_n_smoothing
//     total = np.sum(c2, axis=1): float32, pairwise -- runs of <= 128 elements with 8 interleaved accumulators combined as
//             ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), halves split on multiples of 8 (loops_utils.h.src, *_pairwise_sum); a row has at
//             most 8192 elements here (T = 2^(N+1) - 1, N <= 12), i.e. one block of NumPy's reduction buffer
//     f     = c / total;   model = -log2(f);   len = float32(level) + model     (quantizer.py:171-175)
// A binder without NumPy passes a loop around its own log2f; the results then follow that library.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vbq_common.h"

namespace vbq {
namespace {

float pairwise_f32(const float *a, int64_t n) {
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; ++i) res = res + a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_f32(a, n2) + pairwise_f32(a + n2, n - n2);
}

void libm_log2_loop(char **args, const intptr_t *dimensions, const intptr_t *steps, void *) {
    const char *in = args[0];
    char *out = args[1];
    for (intptr_t i = 0; i < dimensions[0]; ++i, in += steps[0], out += steps[1])
        *reinterpret_cast<float *>(out) = log2f(*reinterpret_cast<const float *>(in));
}

}  // namespace
}  // namespace vbq

extern "C" int vbq_host_neg_log2_freq_f32(const void *h_counts, int32_t counts_are_i32, int64_t n_rows, int64_t K,
                                          float add_n_smoothing, int32_t add_level, vbq_f32_loop log2_loop, void *log2_data,
                                          float *h_out_model, float *h_out_len) {
    using namespace vbq;
    VBQ_REQUIRE(n_rows >= 0 && K >= 1 && K <= 8192, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_host_neg_log2_freq_f32: bad sizes (a histogram row holds 1 .. 8192 bins)");
    if (n_rows == 0) return VBQ_OK;
    VBQ_REQUIRE(h_counts && (h_out_model || h_out_len), VBQ_ERR_INVALID_ARGUMENT, "vbq_host_neg_log2_freq_f32: null pointer argument");
    if (!log2_loop) log2_loop = libm_log2_loop;
    float row[8192], lg[8192];
    for (int64_t r = 0; r < n_rows; ++r) {
        if (counts_are_i32) {
            const int32_t *c = static_cast<const int32_t *>(h_counts) + r * K;
            for (int64_t k = 0; k < K; ++k) row[k] = (float)c[k] + add_n_smoothing;
        } else {
            const int64_t *c = static_cast<const int64_t *>(h_counts) + r * K;
            for (int64_t k = 0; k < K; ++k) row[k] = (float)c[k] + add_n_smoothing;
        }
        const float total = pairwise_f32(row, K);
        for (int64_t k = 0; k < K; ++k) row[k] = row[k] / total;
        char *args[2] = {reinterpret_cast<char *>(row), reinterpret_cast<char *>(lg)};
        const intptr_t dims[1] = {(intptr_t)K}, steps[2] = {4, 4};
        log2_loop(args, dims, steps, log2_data);
        for (int64_t k = 0; k < K; ++k) {
            const float m = -lg[k];
            if (h_out_model) h_out_model[r * K + k] = m;
            if (h_out_len) h_out_len[r * K + k] = add_level ? (float)k + m : m;
        }
    }
    return VBQ_OK;
}

extern "C" void vbq_host_stage_run(void *stage) {
    vbq_host_stage *s = static_cast<vbq_host_stage *>(stage);
    if (!s) return;
    s->status = vbq_host_neg_log2_freq_f32(s->h_counts, s->counts_are_i32, s->n_rows, s->K, s->add_n_smoothing, s->add_level,
                                           s->log2_loop, s->log2_data, s->h_out_model, s->h_out_len);
    s->runs += 1;
}

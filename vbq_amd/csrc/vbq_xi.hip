// The reference's first, xi-space encoder (img-compression/utils.py:215-304): quantise in the unit interval of the
// prior's CDF instead of on a code-point table.  Two fixed pieces of arithmetic surround three caller-supplied
// functions (squash, unsquash, fun), and those two pieces are what runs here, in the reference's float64:
//   k_xi_intervals   utils.get_all_N_bit_intervals (the numba kernel, :215-260): for every coordinate and every bit
//                    budget n the two n-bit grid points around xi -- the literal bit-by-bit truncation loop, operation
//                    for operation (x - offset may round; the loop's subtractions are then exact)
//   k_xi_select      utils.encode_vectorized after the callables (:291-303): the better endpoint per budget (first
//                    maximum), the rate term lamb * n, the best budget (first maximum), and the gathers
// Golden vectors g3 / g4 (the reference's own functions run in the build container) pin both.   gfx950 / ROCm only.
#include "vbq_common.h"

namespace vbq {
namespace {

__global__ void __launch_bounds__(256)
k_xi_intervals(const double *__restrict__ x, long K, int N, double *__restrict__ left, double *__restrict__ right) {
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += (long)gridDim.x * blockDim.x) {
        const double xk = x[k];
        double width = 1.0;                                   // 2 ** (-n), exact
        for (int n = 0; n <= N; ++n, width *= 0.5) {
            double l, r;
            if (n == 0) {
                l = r = 0.5;                                  // special case: the interval does not contain x (:229-233)
            } else {
                const double offset = width * 0.5;
                const double lo = offset, hi = __dsub_rn(1.0, offset);
                if (xk < lo) {
                    l = r = lo;
                } else if (xk > hi) {
                    l = r = hi;
                } else {
                    const double shifted = __dsub_rn(xk, offset);
                    double rem = shifted, bw = 0.5;
                    for (int i = 1; i <= n; ++i, bw *= 0.5) {
                        const double diff = __dsub_rn(rem, bw);
                        if (diff >= 0.0) rem = diff;
                    }
                    const double x_hat = __dsub_rn(shifted, rem);
                    l = __dadd_rn(x_hat, offset);
                    r = __dadd_rn(l, width);
                }
            }
            left[(long)n * K + k] = l;
            right[(long)n * K + k] = r;
        }
    }
}

// F, ends, unsq: [2][N+1][K] (left endpoints first), as np.stack([left, right]) lays them out (:286-288)
__global__ void __launch_bounds__(256)
k_xi_select(const double *__restrict__ F, const double *__restrict__ ends, const double *__restrict__ unsq, long K, int N,
            double lamb, double *__restrict__ z_hat, long long *__restrict__ num_bits, double *__restrict__ xi_hat,
            double *__restrict__ f_z_hat) {
    const long plane = (long)(N + 1) * K;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += (long)gridDim.x * blockDim.x) {
        double best = 0.0;
        int best_n = 0;
        long best_off = 0;
        for (int n = 0; n <= N; ++n) {
            const long o = (long)n * K + k;
            const double fl = F[o], fr = F[plane + o];
            const bool pick_r = fr > fl;                      // np.argmax over the pair: the first maximum
            const double fm = pick_r ? fr : fl;
            const double reg = __dsub_rn(fm, __dmul_rn(lamb, (double)n));
            if (n == 0 || reg > best) { best = reg; best_n = n; best_off = (pick_r ? plane : 0) + o; }
        }
        z_hat[k] = unsq[best_off];
        xi_hat[k] = ends[best_off];
        num_bits[k] = best_n;
        f_z_hat[k] = best;
    }
}

}  // namespace
}  // namespace vbq

extern "C" int vbq_xi_intervals_f64(const double *d_xi, int64_t K, int32_t N, double *d_left, double *d_right, void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(K >= 0 && N >= 0 && N <= 52, VBQ_ERR_INVALID_ARGUMENT, "vbq_xi_intervals_f64: bad sizes K=%lld N=%d", (long long)K, N);
    if (K == 0) return VBQ_OK;
    VBQ_REQUIRE(d_xi && d_left && d_right, VBQ_ERR_INVALID_ARGUMENT, "vbq_xi_intervals_f64: null pointer argument");
    int64_t gx = (K + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_xi_intervals, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_xi, (long)K, (int)N,
                       d_left, d_right);
    VBQ_CHECK_LAUNCH("xi_intervals");
    return VBQ_OK;
}

extern "C" int vbq_xi_select_f64(const double *d_F, const double *d_endpoints, const double *d_unsquashed, int64_t K, int32_t N,
                                 double lamb, double *d_z_hat, int64_t *d_num_bits, double *d_xi_hat, double *d_f_z_hat,
                                 void *stream) {
    using namespace vbq;
    VBQ_REQUIRE(K >= 0 && N >= 0 && N <= 52, VBQ_ERR_INVALID_ARGUMENT, "vbq_xi_select_f64: bad sizes K=%lld N=%d", (long long)K, N);
    if (K == 0) return VBQ_OK;
    VBQ_REQUIRE(d_F && d_endpoints && d_unsquashed && d_z_hat && d_num_bits && d_xi_hat && d_f_z_hat, VBQ_ERR_INVALID_ARGUMENT,
                "vbq_xi_select_f64: null pointer argument");
    int64_t gx = (K + 255) / 256;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(k_xi_select, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_F, d_endpoints,
                       d_unsquashed, (long)K, (int)N, lamb, d_z_hat, reinterpret_cast<long long *>(d_num_bits), d_xi_hat, d_f_z_hat);
    VBQ_CHECK_LAUNCH("xi_select");
    return VBQ_OK;
}

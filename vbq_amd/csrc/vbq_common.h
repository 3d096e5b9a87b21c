// Shared device/host helpers for the VBQ HIP kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vbq.h"

namespace vbq {

void set_error(const char *fmt, ...);

// Compute units of the current device (hipDeviceProp_t::multiProcessorCount, cached per device): what the persistent
// grids are sized by.  256 on MI355X; never hard-coded.
int num_cus();
// Workgroup slots of a resident grid with `per_cu` workgroups per CU, minus the `reserved` slots the CALLER leaves to a kernel of
// another stream (the `reserved_workgroups` argument of the entry points: the distributed pipeline's overlapped all-reduce),
// never below one per CU.  default_reserved_workgroups(): what a negative argument means -- VBQ_RESERVED_WORKGROUPS, read once.
int64_t resident_slots(int per_cu, int reserved);
int default_reserved_workgroups();

#define VBQ_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            ::vbq::set_error(__VA_ARGS__);      \
            return (code);                      \
        }                                       \
    } while (0)

#define VBQ_CHECK_LAUNCH(what)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            ::vbq::set_error("%s: %s", (what), hipGetErrorString(e__));               \
            return VBQ_ERR_LAUNCH;                                                    \
        }                                                                             \
    } while (0)

// Issue priority in resident ("persistent") grids.  The SIMD arbitrates instruction issue between its waves by priority, then
// by AGE: with every wave at priority 0 the oldest one runs nearly unimpeded and the youngest gets the leftover slots.  In a grid
// whose workgroups are all resident from start to end that order never changes: the oldest wave of a SIMD finishes early, the
// youngest is left to run alone at a fraction of the issue rate (K1 on Kodak-24: 394 us with 4 workgroups per CU x 18
// iterations, against 358 us for a dynamic grid of 18 x 4 although that one has a half-empty last round).  Rotating the
// priority over the resident waves -- wave slot + iteration, modulo 4 -- lets them progress at the same pace and finish
// together: K1t 138 -> 122 us, K1 394 -> 335 us (EXPERIMENTS.md, "issue priority").  All four levels are used: two waves on one
// level are ordered by age again.  s_setprio takes an immediate.
__device__ __forceinline__ void set_issue_priority(unsigned int p) {
    p = __builtin_amdgcn_readfirstlane(p) & 3u;
    if (p == 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}
// The wave's slot on its SIMD (HW_ID[3:0]): distinct for the waves that compete for one SIMD's issue cycles.
__device__ __forceinline__ unsigned int wave_slot() { return __builtin_amdgcn_s_getreg((3 << 11) | 4); }
// One step of the rotation, at the top of a kernel's element loop: `rot` starts at wave_slot() and walks 3, 2, 1, 0, 3, ...
// (-DVBQ_NO_ISSUE_ROTATION builds the kernels without it, for A/B timing).
__device__ __forceinline__ void rotate_issue_priority(unsigned int &rot) {
#ifndef VBQ_NO_ISSUE_ROTATION
    set_issue_priority(rot);
    rot += 3u;
#endif
}

constexpr int kWave = 64;
constexpr int kTileChannels = 16;   // channel tables resident in LDS per workgroup (tiled kernel)
constexpr int kMaxLambdaChunk = 32; // lambdas handled per launch (penalty table staged in LDS)

__host__ __device__ constexpr int table_size(int N) { return (2 << N) - 1; }

// Exactly-rounded f32 pieces of utils.py:319-320:  -0.5 * ((P - mu) / sigma) ** 2
// (four separately rounded operations; the file is also built with -ffp-contract=off).
__device__ __forceinline__ float neg_half_sq_err(float P, float mu, float sigma) {
    float d = __fsub_rn(P, mu);
    float t = __fdiv_rn(d, sigma);
    float q = __fmul_rn(t, t);
    return __fmul_rn(-0.5f, q);
}

// Bit depths the kernels are instantiated for.  The reference uses N = 10 only (post_process.py:117); the others are
// an extension that costs build time: -DVBQ_ONLY_N10 builds the reference's depth alone.
#ifdef VBQ_ONLY_N10
#define VBQ_FOR_EACH_N(X) X(10)
#else
#define VBQ_FOR_EACH_N(X) X(12) X(11) X(10) X(9) X(8) X(7) X(6) X(5) X(4)
#endif

// The lambdas of one launch rounded to f32 (TF casts the Python scalar to the tensor dtype), passed in the kernel arguments:
// the fast kernels form their penalties fl32(lambda) * len in the prologue instead of reading a table another kernel prepared.
struct Lambdas32 {
    float lam[kMaxLambdaChunk];
};

// vbq_quantize_fast.hip.  Elements of channel c start at c * ch_stride (n_per_ch of them are processed); E is the
// distance between the lambda planes of the outputs.  level_counts != NULL selects the counting mode (no element
// output).  wg_per_cu in 1..5 makes the grid persistent at that many workgroups per CU (0: the default sizing); `reserved`
// workgroup slots are left to a kernel of another stream.
template <int N>
int launch_quant_fast(const float *mu, const float *sg, int64_t n_per_ch, int64_t ch_stride, int32_t n_ch, const float *table,
                      const Lambdas32 &lam, const float *len, int32_t L, uint16_t *out_idx, float *out_zhat,
                      float *out_bits, int64_t E, int vec_ok,
                      unsigned long long *level_counts, int wg_per_cu, int reserved, hipStream_t st);

// K1p (vbq_quantize_fast.hip): one to four lambdas per call with exact pruning of the descent; indices only.  Returns 1 when the
// call is not of that kind (the caller takes launch_quant_fast).  Valid for ANY penalties (literal comparisons only).
template <int N>
int launch_quant_pruned(const float *mu, const float *sg, int64_t n_per_ch, int64_t ch_stride, int32_t n_ch, const float *table,
                        const Lambdas32 &lam, const float *len, int32_t L, uint16_t *out_idx, int64_t E, int vec_ok, hipStream_t st);

// K1t (vbq_quantize_fast.hip): first entropy-model pass without a per-lambda loop; N = 10, raw lengths.  Returns 1 when the
// lambda sweep is not eligible (caller falls back to the dense counting kernel).
int launch_level_counts_hull10(const float *mu, const float *sg, int64_t n_per_ch, int64_t ch_stride, int32_t n_ch,
                               const float *table, const double *lam, int32_t L, int vec_ok,
                               unsigned long long *level_counts, int reserved, hipStream_t st);

// K1e (vbq_quantize_fast.hip): rank indices of a raw-length lambda sweep from K1t's thresholds; N = 10.  Returns 1 when the
// sweep is not eligible (caller falls back to launch_quant_fast).
int launch_quant_hull_idx10(const float *mu, const float *sg, int64_t n_per_ch, int64_t ch_stride, int32_t n_ch,
                            const float *table, const double *lam, int32_t L, int vec_ok, uint16_t *out_idx, int64_t E,
                            hipStream_t st);

}  // namespace vbq

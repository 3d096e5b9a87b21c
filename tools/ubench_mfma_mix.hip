// Does an f32 MFMA co-issue with VALU work?  (developer tool, not product code)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_mix.hip -o tools/bin/ubench_mfma_mix && tools/bin/ubench_mfma_mix
// Per iteration: 32 independent v_add_f32 (8 chains x 4) and M v_mfma_f32_32x32x2_f32 (4 accumulators), M = 0, 1, 2, 4.
// Prints SIMD cycles per iteration at 4 waves per SIMD (clock from wall time is approximate).
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int M>
__global__ void __launch_bounds__(256) k_mix(float *out, int iters) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x + i;
    const float s = out[0];
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
            if (rep < M) acc[rep] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[0], s, acc[rep], 0, 0, 0);
        }
    }
    float t = 0;
    for (int i = 0; i < 8; ++i) t += r[i];
    for (int a = 0; a < 4; ++a) t += acc[a][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int M>
void run(float *d, const char *name) {
    const int iters = 4096, blocks = 256 * 4;                     // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_mix<M>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_mix<M>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.4e9 / iters / 4.0;            // per iteration per wave, at a nominal 2.4 GHz
    printf("%-28s %8.3f ms  ~%6.1f SIMD cycles per (32 adds + %d MFMA) of one wave\n", name, ms, cyc, M);
}

int main() {
    float *d;
    hipMalloc(&d, sizeof(float) * 256 * 1024 * 4);
    hipMemset(d, 0, sizeof(float) * 256 * 1024 * 4);
    run<0>(d, "32 v_add_f32");
    run<1>(d, "32 v_add_f32 + 1 mfma f32");
    run<2>(d, "32 v_add_f32 + 2 mfma f32");
    run<4>(d, "32 v_add_f32 + 4 mfma f32");
    return 0;
}

// f64 / conversion throughput (developer tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define DEFK(NAME, ASM)                                                                   \
    __global__ void __launch_bounds__(256) k_##NAME(double *out, int iters) {             \
        double r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;                    \
        double s = out[0] + 1.0000001;                                                    \
        for (int i = 0; i < iters; ++i) {                                                 \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(0) ASM(1) ASM(2) ASM(3)          \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(0) ASM(1) ASM(2) ASM(3)          \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(s));              \
        }                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;                    \
    }
#define A_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %4\n"
#define A_ADD64(i) "v_add_f64 %" #i ", %" #i ", %4\n"
#define A_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %4, %4\n"
#define A_CVT6432(i) "v_cvt_f32_f64 %" #i ", %" #i "\n"
#define A_CVT3264(i) "v_cvt_f64_f32 %" #i ", %" #i "\n"
#define A_RCP64(i) "v_rcp_f64 %" #i ", %" #i "\n"
DEFK(mul64, A_MUL64)
DEFK(add64, A_ADD64)
DEFK(fma64, A_FMA64)
DEFK(cvt6432, A_CVT6432)
DEFK(cvt3264, A_CVT3264)
DEFK(rcp64, A_RCP64)
typedef void (*kfn)(double *, int);
struct Entry { const char *name; kfn fn; };
int main() {
    Entry ks[] = {{"v_mul_f64", k_mul64}, {"v_add_f64", k_add64}, {"v_fma_f64", k_fma64}, {"v_cvt_f32_f64", k_cvt6432},
                  {"v_cvt_f64_f32", k_cvt3264}, {"v_rcp_f64", k_rcp64}};
    double *out;
    (void)hipMalloc(&out, 256 * 8 * 4 * 256 * sizeof(double));
    (void)hipMemset(out, 0, 4096);
    const int iters = 16384, wps = 4;
    for (auto &e : ks) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(e.fn, dim3(256 * wps), dim3(256), 0, 0, out, 64);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(e.fn, dim3(256 * wps), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        printf("%-20s %8.3f ms  %6.2f cycles/wave-instr/SIMD at 2.4 GHz\n", e.name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps));
    }
    return 0;
}

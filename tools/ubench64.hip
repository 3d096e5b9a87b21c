// f64 / conversion throughput (developer tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define DEFK(NAME, ASM)                                                                   \
    __global__ void __launch_bounds__(256) k_##NAME(double *out, int iters) {             \
        double r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;                    \
        float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;                     \
        double s = out[0] + 1.0000001;                                                    \
        for (int i = 0; i < iters; ++i) {                                                 \
            asm volatile(ASM(0, 4) ASM(1, 5) ASM(2, 6) ASM(3, 7) ASM(0, 4) ASM(1, 5) ASM(2, 6) ASM(3, 7)  \
                         ASM(0, 4) ASM(1, 5) ASM(2, 6) ASM(3, 7) ASM(0, 4) ASM(1, 5) ASM(2, 6) ASM(3, 7)  \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(s)); \
        }                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + f0 + f1 + f2 + f3; \
    }
#define A_MUL64(i, j) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define A_ADD64(i, j) "v_add_f64 %" #i ", %" #i ", %8\n"
#define A_FMA64(i, j) "v_fma_f64 %" #i ", %" #i ", %8, %8\n"
#define A_CVT6432(i, j) "v_cvt_f32_f64 %" #j ", %" #i "\n"
#define A_CVT3264(i, j) "v_cvt_f64_f32 %" #i ", %" #j "\n"
#define A_RCP64(i, j) "v_rcp_f64 %" #i ", %" #i "\n"
#define A_PKADD(i, j) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define A_PKMUL(i, j) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define A_PKFMA(i, j) "v_pk_fma_f32 %" #i ", %" #i ", %8, %8\n"
DEFK(mul64, A_MUL64)
DEFK(add64, A_ADD64)
DEFK(fma64, A_FMA64)
DEFK(cvt6432, A_CVT6432)
DEFK(cvt3264, A_CVT3264)
DEFK(rcp64, A_RCP64)
DEFK(pkadd, A_PKADD)
DEFK(pkmul, A_PKMUL)
DEFK(pkfma, A_PKFMA)
typedef void (*kfn)(double *, int);
struct Entry { const char *name; kfn fn; };
int main() {
    Entry ks[] = {{"v_mul_f64", k_mul64}, {"v_add_f64", k_add64}, {"v_fma_f64", k_fma64}, {"v_cvt_f32_f64", k_cvt6432},
                  {"v_cvt_f64_f32", k_cvt3264}, {"v_rcp_f64", k_rcp64}, {"v_pk_add_f32", k_pkadd},
                  {"v_pk_mul_f32", k_pkmul}, {"v_pk_fma_f32", k_pkfma}};
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
        printf("%-20s %8.3f ms  %6.2f cycles/wave-instr/SIMD at 2.4 GHz (v_add_f32 = 2.85)\n", e.name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps));
    }
    return 0;
}

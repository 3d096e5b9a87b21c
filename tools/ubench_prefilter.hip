// Issue rates of the instructions a reduced-precision pre-filter / a 64-bit-key argmin for K1 would be built from
// ("packed bf16/f16 costs to shortlist <= 2 levels").  Developer tool, not product code.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_prefilter.hip -o /tmp/ubp && /tmp/ubp
// gfx950 has NO packed bf16 arithmetic (v_pk_add_bf16 / v_pk_min_bf16 do not assemble), so bf16 is out before any timing.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define DEFK(NAME, ASM)                                                                   \
    __global__ void __launch_bounds__(256) k_##NAME(double *out, int iters) {             \
        double r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;                    \
        float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;                     \
        double s = out[0] + 1.0000001;                                                    \
        float sf = (float)s;                                                              \
        for (int i = 0; i < iters; ++i) {                                                 \
            asm volatile(ASM(0, 4) ASM(1, 5) ASM(2, 6) ASM(3, 7) ASM(0, 4) ASM(1, 5) ASM(2, 6) ASM(3, 7)  \
                         ASM(0, 4) ASM(1, 5) ASM(2, 6) ASM(3, 7) ASM(0, 4) ASM(1, 5) ASM(2, 6) ASM(3, 7)  \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(s), "v"(sf) : "vcc", "s10", "s11"); \
        }                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + f0 + f1 + f2 + f3; \
    }
#define A_ADD32(i, j) "v_add_f32 %" #j ", %" #j ", %9\n"
#define A_MIN364(i, j) "v_min3_f32 %" #j ", %" #j ", %9, %9\n"
#define A_PKADD16(i, j) "v_pk_add_f16 %" #j ", %" #j ", %9\n"
#define A_PKMIN16(i, j) "v_pk_min_f16 %" #j ", %" #j ", %9\n"
#define A_PKMIN316(i, j) "v_pk_minimum3_f16 %" #j ", %" #j ", %9, %9\n"
#define A_PKMUL16(i, j) "v_pk_mul_f16 %" #j ", %" #j ", %9\n"
#define A_CVTPK(i, j) "v_cvt_pkrtz_f16_f32 %" #j ", %" #j ", %9\n"
#define A_PKMINU16(i, j) "v_pk_min_u16 %" #j ", %" #j ", %9\n"
#define A_MIN64(i, j) "v_min_f64 %" #i ", %" #i ", %8\n"
#define A_MAX64(i, j) "v_max_f64 %" #i ", %" #i ", %8\n"
#define A_CMPEQ(i, j) "v_cmp_eq_u32 s[10:11], %" #j ", %9\n"
#define A_CMPEQV(i, j) "v_cmp_eq_u32 vcc, %" #j ", %9\n"
DEFK(add32, A_ADD32)
DEFK(min3, A_MIN364)
DEFK(pkadd16, A_PKADD16)
DEFK(pkmin16, A_PKMIN16)
DEFK(pkmin316, A_PKMIN316)
DEFK(pkmul16, A_PKMUL16)
DEFK(cvtpk, A_CVTPK)
DEFK(pkminu16, A_PKMINU16)
DEFK(min64, A_MIN64)
DEFK(max64, A_MAX64)
DEFK(cmpeq, A_CMPEQ)
DEFK(cmpeqv, A_CMPEQV)
typedef void (*kfn)(double *, int);
struct Entry { const char *name; kfn fn; };
int main() {
    Entry ks[] = {{"v_add_f32 (reference)", k_add32}, {"v_min3_f32 (reference)", k_min3}, {"v_pk_add_f16", k_pkadd16},
                  {"v_pk_min_f16", k_pkmin16}, {"v_pk_minimum3_f16", k_pkmin316}, {"v_pk_mul_f16", k_pkmul16},
                  {"v_cvt_pkrtz_f16_f32", k_cvtpk}, {"v_pk_min_u16", k_pkminu16}, {"v_min_f64", k_min64}, {"v_max_f64", k_max64},
                  {"v_cmp_eq_u32 -> sgpr", k_cmpeq}, {"v_cmp_eq_u32 -> vcc", k_cmpeqv}};
    double *out;
    (void)hipMalloc(&out, 256 * 8 * 4 * 256 * sizeof(double));
    (void)hipMemset(out, 0, 4096);
    const int iters = 16384, wps = 4;
    double base = 0;
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
        if (base == 0) base = ms;
        printf("%-26s %8.3f ms  = %5.2f x v_add_f32   (%5.2f cycles / wave-instr / SIMD if v_add_f32 = 2)\n", e.name, ms, ms / base, 2.0 * ms / base);
    }
    return 0;
}

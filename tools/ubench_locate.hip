// K1's "locate" per level, three encodings, as issue cost per (element, level) pair on gfx950 (developer tool).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_locate.hip -o tools/bin/ubench_locate && tools/bin/ubench_locate
// Every kernel runs 8 independent mask chains ne_k = 2 ne_k + [c_k != S] per thread, 16 pairs per loop body.
//   A  v_sub_f32 t, S, c ; v_alignbit_b32 ne, ne, t, 31            (what K1 ships: sign of S - c shifted in)
//   B  v_sub_co_u32 t, vcc, S, c ; v_addc_co_u32 ne, vcc, ne, ne, vcc   (borrow of the bit patterns, carried in)
//   C  v_cmp_lt_f32 vcc, S, c ; v_addc_co_u32 ne, vcc, ne, ne, vcc
//   D  v_cmp_lt_u32 vcc, S, c ; v_addc_co_u32 ne, vcc, ne, ne, vcc
// and the single instructions v_sub_co_u32 / v_subb_co_u32 / v_addc_co_u32 / v_add_co_u32.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define DEFK(NAME, ASM)                                                                                             \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned *out, int iters) {                                     \
        unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6,      \
                 r7 = r0 + 7;                                                                                       \
        unsigned s = out[0], c = out[1] + threadIdx.x, t;                                                           \
        for (int i = 0; i < iters; ++i) {                                                                           \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                    \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                    \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "=&v"(t) \
                         : "v"(s), "v"(c) : "vcc");                                                                 \
        }                                                                                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + t;                     \
    }

// %0..%7 chains, %8 scratch, %9 = S, %10 = c
#define P_A(i) "v_sub_f32 %8, %9, %10\n v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define P_B(i) "v_sub_co_u32 %8, vcc, %9, %10\n v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n"
#define P_C(i) "v_cmp_lt_f32 vcc, %9, %10\n v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n"
#define P_D(i) "v_cmp_lt_u32 vcc, %9, %10\n v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n"
#define S_SUBCO(i) "v_sub_co_u32 %" #i ", vcc, %9, %" #i "\n"
#define S_SUBB(i) "v_subb_co_u32 %" #i ", vcc, %9, %" #i ", vcc\n"
#define S_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n"
#define S_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %9\n"
#define S_SUBF(i) "v_sub_f32 %" #i ", %9, %" #i "\n"
#define S_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %9, 31\n"
#define S_CMPF(i) "v_cmp_lt_f32 vcc, %" #i ", %9\n"
#define S_CMPU(i) "v_cmp_lt_u32 vcc, %" #i ", %9\n"
// B with the two halves of neighbouring chains interleaved is impossible (one VCC); what CAN sit between them is other
// VALU work that does not touch VCC: a v_add_f32 after every pair, as the cost adds of the next lambda would
#define P_BX(i) "v_sub_co_u32 %8, vcc, %9, %10\n v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n v_add_f32 %8, %9, %10\n"
#define P_AX(i) "v_sub_f32 %8, %9, %10\n v_alignbit_b32 %" #i ", %" #i ", %8, 31\n v_add_f32 %8, %9, %10\n"

DEFK(pA, P_A) DEFK(pB, P_B) DEFK(pC, P_C) DEFK(pD, P_D) DEFK(pAX, P_AX) DEFK(pBX, P_BX)
DEFK(subco, S_SUBCO) DEFK(subb, S_SUBB) DEFK(addc, S_ADDC) DEFK(addco, S_ADDCO)
DEFK(subf, S_SUBF) DEFK(align, S_ALIGN) DEFK(cmpf, S_CMPF) DEFK(cmpu, S_CMPU)

typedef void (*kfn)(unsigned *, int);
struct Entry { const char *name; kfn fn; int instr; };

int main() {
    Entry ks[] = {{"A sub_f32 + alignbit", k_pA, 2}, {"B sub_co_u32 + addc_co_u32", k_pB, 2}, {"C cmp_lt_f32 + addc", k_pC, 2},
                  {"D cmp_lt_u32 + addc", k_pD, 2}, {"A + v_add_f32", k_pAX, 3}, {"B + v_add_f32", k_pBX, 3},
                  {"v_sub_co_u32", k_subco, 1}, {"v_subb_co_u32", k_subb, 1}, {"v_addc_co_u32", k_addc, 1},
                  {"v_add_co_u32", k_addco, 1}, {"v_sub_f32", k_subf, 1}, {"v_alignbit_b32", k_align, 1},
                  {"v_cmp_lt_f32 vcc", k_cmpf, 1}, {"v_cmp_lt_u32 vcc", k_cmpu, 1}};
    unsigned *out;
    hipMalloc(&out, 256 * 8 * 4 * 256 * sizeof(unsigned));
    hipMemset(out, 0, 4096);
    const int iters = 16384;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double clk = p.clockRate * 1e3;
    for (int wps = 1; wps <= 4; wps += 3) {
        printf("--- %d wave(s) per SIMD (nominal clock %.2f GHz): cycles per GROUP (pair / triple / single) per wave per SIMD\n", wps, clk / 1e9);
        for (auto &e : ks) {
            const int blocks = 256 * wps;
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 64);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, iters);       // clock ramp
            hipEventRecord(a);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double groups_per_simd = (double)iters * 16 * wps;
            printf("%-30s %8.3f ms  %6.2f cycles / group (%d instr)\n", e.name, ms, ms * 1e-3 * clk / groups_per_simd, e.instr);
        }
    }
    return 0;
}

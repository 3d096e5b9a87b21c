// Instruction-throughput micro-benchmark for gfx950 (developer tool, not product code).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
// Each kernel issues 8 independent chains of one instruction, 2048 iterations, on every SIMD
// with WAVES waves per SIMD; prints SIMD-cycles per wave-instruction relative to wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFK(NAME, ASM)                                                                   \
    __global__ void __launch_bounds__(256) k_##NAME(float *out, int iters) {              \
        float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, \
              r7 = r0 + 7;                                                                \
        float s = out[0];                                                                 \
        for (int i = 0; i < iters; ++i) {                                                 \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)          \
                         ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)          \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                         : "v"(s));                                                       \
        }                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; \
    }

#define A_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define A_SUB(i) "v_sub_f32 %" #i ", %8, %" #i "\n"
#define A_MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define A_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %8\n"
#define A_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CMP(i) "v_cmp_gt_f32 vcc, %" #i ", %8\n"
#define A_CMPE64(i) "v_cmp_gt_f32 s[10:11], %" #i ", %8\n"
#define A_FFBH(i) "v_ffbh_u32 %" #i ", %" #i "\n"
#define A_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 11, 11\n"
#define A_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define A_MINU(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define A_MIN3U(i) "v_min3_u32 %" #i ", %" #i ", %8, %8\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %8\n"
#define A_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define A_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %8\n"
#define A_ADDS(i) "v_add_f32 %" #i ", s8, %" #i "\n"

#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define A_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define A_SUBU(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define A_FMAC(i) "v_fmac_f32 %" #i ", %8, %8\n"
#define A_MADU24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define A_MULU24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define A_CVTI(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define A_ADDLIT(i) "v_add_f32 %" #i ", 0x3fc00000, %" #i "\n"
#define A_ADDINL(i) "v_add_f32 %" #i ", 1.0, %" #i "\n"
#define A_CND2(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[10:11]\n"
#define A_DIVFIX(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %8\n"
#define A_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %8\n"
#define A_DIVFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %8\n"
#define A_SUBREV(i) "v_subrev_f32 %" #i ", %" #i ", %8\n"
#define A_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define A_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %" #i ", %8\n"
#define A_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define A_MIN3I(i) "v_min3_i32 %" #i ", %" #i ", %8, %8\n"
#define A_MAXU(i) "v_max_u32 %" #i ", %" #i ", %8\n"
#define A_ASHR(i) "v_ashrrev_i32 %" #i ", 31, %" #i "\n"

DEFK(lshr, A_LSHR)
DEFK(lshl, A_LSHL)
DEFK(or_, A_OR)
DEFK(xor_, A_XOR)
DEFK(max, A_MAX)
DEFK(mov, A_MOV)
DEFK(addco, A_ADDCO)
DEFK(addc, A_ADDC)
DEFK(subu, A_SUBU)
DEFK(fmac, A_FMAC)
DEFK(madu24, A_MADU24)
DEFK(mulu24, A_MULU24)
DEFK(cvti, A_CVTI)
DEFK(addlit, A_ADDLIT)
DEFK(addinl, A_ADDINL)
DEFK(cnd2, A_CND2)
DEFK(divfix, A_DIVFIX)
DEFK(divscale, A_DIVSCALE)
DEFK(divfmas, A_DIVFMAS)
DEFK(subrev, A_SUBREV)
DEFK(bcnt, A_BCNT)
DEFK(lshladd, A_LSHLADD)
DEFK(sqrt, A_SQRT)
DEFK(min3i, A_MIN3I)
DEFK(maxu, A_MAXU)
DEFK(ashr, A_ASHR)
DEFK(add, A_ADD)
DEFK(fma, A_FMA)
DEFK(sub, A_SUB)
DEFK(min, A_MIN)
DEFK(min3, A_MIN3)
DEFK(alignbit, A_ALIGN)
DEFK(and_, A_AND)
DEFK(lshlor, A_LSHLOR)
DEFK(cndmask, A_CNDMASK)
DEFK(cmp, A_CMP)
DEFK(cmpe64, A_CMPE64)
DEFK(ffbh, A_FFBH)
DEFK(mul, A_MUL)
DEFK(rcp, A_RCP)
DEFK(bfe, A_BFE)
DEFK(addu, A_ADDU)
DEFK(add3, A_ADD3)
DEFK(minu, A_MINU)
DEFK(min3u, A_MIN3U)
DEFK(andor, A_ANDOR)
DEFK(med3, A_MED3)
DEFK(adds, A_ADDS)

typedef void (*kfn)(float *, int);
struct Entry { const char *name; kfn fn; };

int main(int argc, char **argv) {
    Entry ks[] = {{"v_add_f32", k_add}, {"v_fma_f32", k_fma}, {"v_sub_f32", k_sub}, {"v_min_f32", k_min},
                  {"v_min3_f32", k_min3}, {"v_alignbit_b32", k_alignbit}, {"v_and_b32", k_and_},
                  {"v_lshl_or_b32", k_lshlor}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_gt_f32 vcc", k_cmp},
                  {"v_cmp_gt_f32 sgpr", k_cmpe64}, {"v_ffbh_u32", k_ffbh}, {"v_mul_f32", k_mul}, {"v_rcp_f32", k_rcp},
                  {"v_bfe_u32", k_bfe}, {"v_add_u32", k_addu}, {"v_add3_u32", k_add3}, {"v_min_u32", k_minu},
                  {"v_min3_u32", k_min3u}, {"v_and_or_b32", k_andor}, {"v_med3_f32", k_med3},
                  {"v_add_f32 sgpr", k_adds}, {"v_lshrrev_b32", k_lshr}, {"v_lshlrev_b32", k_lshl}, {"v_or_b32", k_or_},
                  {"v_xor_b32", k_xor_}, {"v_max_f32", k_max}, {"v_mov_b32", k_mov}, {"v_add_co_u32", k_addco},
                  {"v_addc_co_u32", k_addc}, {"v_sub_u32", k_subu}, {"v_fmac_f32", k_fmac}, {"v_mad_u32_u24", k_madu24},
                  {"v_mul_u32_u24", k_mulu24}, {"v_cvt_f32_i32", k_cvti}, {"v_add_f32 literal", k_addlit},
                  {"v_add_f32 inline1.0", k_addinl}, {"v_cndmask sgprmask", k_cnd2}, {"v_div_fixup_f32", k_divfix},
                  {"v_div_scale_f32", k_divscale}, {"v_div_fmas_f32", k_divfmas}, {"v_subrev_f32", k_subrev},
                  {"v_bcnt_u32_b32", k_bcnt}, {"v_lshl_add_u32", k_lshladd}, {"v_sqrt_f32", k_sqrt},
                  {"v_min3_i32", k_min3i}, {"v_max_u32", k_maxu}, {"v_ashrrev_i32", k_ashr}};
    float *out;
    hipMalloc(&out, 256 * 8 * 4 * 256 * sizeof(float));
    hipMemset(out, 0, 4096);
    const int iters = 16384;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double clk = p.clockRate * 1e3;   // Hz (nominal)
    for (int wps = 3; wps <= 4; wps += 1) {
        printf("--- %d wave(s) per SIMD (nominal clock %.2f GHz)\n", wps, clk / 1e9);
        for (auto &e : ks) {
            const int blocks = 256 * wps;     // 256 threads = 4 waves = one per SIMD of a CU
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 64);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double instr_per_simd = (double)iters * 16 * wps;
            printf("%-20s %8.3f ms  %6.2f cycles / wave-instr / SIMD (at nominal clock)\n", e.name, ms,
                   ms * 1e-3 * clk / instr_per_simd);
        }
    }
    return 0;
}

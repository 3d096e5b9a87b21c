// Developer tool: a kernel that only OCCUPIES workgroup slots -- the footprint of a collective's kernel (RCCL all-reduce:
// a few dozen workgroups of 256-512 threads with >= 128 VGPRs, resident for the whole transfer) beside the resident grids
// of K1t / K1 (tools/resident_vs_collective.py).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/spin.hip -o tools/bin/libspin.so
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_spin(uint64_t ticks, uint32_t *sink) {
    // v127 clobbered: the allocation is 128 VGPRs per lane, an RCCL-class footprint (two such waves per SIMD leave room for
    // two of K1's four 112-VGPR waves)
    asm volatile("v_mov_b32 v127, 0" ::: "v127");
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();           // 100 MHz
    uint32_t n = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(8);                                 // occupies slots, not issue cycles (a collective mostly waits)
        ++n;
    }
    if (sink && n == 0xffffffffu) sink[0] = n;
}

// One wave, a handful of registers: a co-resident kernel that takes (almost) no slots -- separates "a second queue is busy"
// from "slots are taken".
__global__ void __launch_bounds__(64) k_spin_small(uint64_t ticks, uint32_t *sink) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t n = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        ++n;
    }
    if (sink && n == 0xffffffffu) sink[0] = n;
}

// The same without the clock: a counted loop of s_sleep (mode 1) or of integer adds (mode 2) -- rules the s_memrealtime polling
// in or out as the cause of what the neighbours lose.
__global__ void __launch_bounds__(64) k_spin_counted(uint64_t iters, int mode, uint32_t *sink) {
    uint32_t n = threadIdx.x;
    for (uint64_t i = 0; i < iters; ++i) {
        if (mode == 1) __builtin_amdgcn_s_sleep(8);
        else asm volatile("v_add_u32 %0, %0, %0" : "+v"(n));
    }
    if (sink && n == 0xfffffffeu) sink[0] = n;
}

extern "C" int spin_launch_counted(void *stream, int n_wg, int mode, double milliseconds) {
    // s_sleep 8 = 512 cycles at ~2.1 GHz = 0.24 us per iteration; the add loop ~8 cycles per iteration
    const uint64_t iters = (uint64_t)(milliseconds * (mode == 1 ? 4000.0 : 250000.0));
    hipLaunchKernelGGL(k_spin_counted, dim3(n_wg), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), iters, mode, (uint32_t *)nullptr);
    return (int)hipGetLastError();
}

extern "C" int spin_launch(void *stream, int n_wg, int threads, double milliseconds) {
    const uint64_t ticks = (uint64_t)(milliseconds * 1e5);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (threads == 64) hipLaunchKernelGGL(k_spin_small, dim3(n_wg), dim3(64), 0, st, ticks, (uint32_t *)nullptr);
    else if (threads == 512) hipLaunchKernelGGL(k_spin<512>, dim3(n_wg), dim3(512), 0, st, ticks, (uint32_t *)nullptr);
    else hipLaunchKernelGGL(k_spin<256>, dim3(n_wg), dim3(256), 0, st, ticks, (uint32_t *)nullptr);
    return (int)hipGetLastError();
}

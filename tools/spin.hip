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

extern "C" int spin_launch(void *stream, int n_wg, int threads, double milliseconds) {
    const uint64_t ticks = (uint64_t)(milliseconds * 1e5);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (threads == 512) hipLaunchKernelGGL(k_spin<512>, dim3(n_wg), dim3(512), 0, st, ticks, (uint32_t *)nullptr);
    else hipLaunchKernelGGL(k_spin<256>, dim3(n_wg), dim3(256), 0, st, ticks, (uint32_t *)nullptr);
    return (int)hipGetLastError();
}

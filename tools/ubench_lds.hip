// LDS atomic / read throughput under different address patterns (developer tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE, bool ATOMIC>
__global__ void __launch_bounds__(256) k(unsigned *out, int iters, int nbins) {
    __shared__ unsigned h[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) h[i] = 0;
    __syncthreads();
    unsigned lane = threadIdx.x & 63, x = threadIdx.x * 2654435761u + blockIdx.x, acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            unsigned a;
            if (MODE == 0) a = lane + 64 * u;                  // conflict-free, distinct
            else if (MODE == 1) a = (lane * 32 + u) & 8191;    // one bank
            else if (MODE == 2) a = u;                         // one address
            else { x = x * 1664525u + 1013904223u; a = (x >> 8) % (unsigned)nbins; }   // random bins
            if (ATOMIC) atomicAdd(&h[a], 1u); else acc += h[a];
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x] + acc;
}
template <int MODE, bool ATOMIC>
void run(const char *name, unsigned *out, int nbins) {
    const int iters = 4096, blocks = 256 * 4;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, ATOMIC>), dim3(blocks), dim3(256), 0, 0, out, 16, nbins);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, ATOMIC>), dim3(blocks), dim3(256), 0, 0, out, iters, nbins);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double winstr_per_cu = (double)iters * 8 * 4 /*waves/block*/ * (blocks / 256.0);
    printf("%-34s %8.3f ms  %7.1f cycles per wave-instr per CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / winstr_per_cu);
}
int main() {
    unsigned *out;
    (void)hipMalloc(&out, 256 * 4 * 256 * 4);
    run<0, true>("atomic add, conflict-free", out, 0);
    run<1, true>("atomic add, one bank (64 addrs)", out, 0);
    run<2, true>("atomic add, one address", out, 0);
    run<3, true>("atomic add, random 2047 bins", out, 2047);
    run<3, true>("atomic add, random 64 bins", out, 64);
    run<3, true>("atomic add, random 8 bins", out, 8);
    run<0, false>("read b32, conflict-free", out, 0);
    run<1, false>("read b32, one bank", out, 0);
    run<3, false>("read b32, random 2047", out, 2047);
    return 0;
}

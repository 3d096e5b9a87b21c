// Write-bandwidth patterns for the index output of the sweep kernels (idx[l][element], one row per sweep point): how fast
// can 2 B x L x n leave the chip when a wave writes 256 B (or 512 B) into each of L rows?  Developer tool, not product code.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_store.hip -o /tmp/ubs && /tmp/ubs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void __launch_bounds__(256) k_linear(uint4 *out, long n16) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) out[i] = make_uint4(i, 1, 2, 3);
}
template <int W>   // W = 32-bit words per lane per row
__global__ void __launch_bounds__(256) k_rows(uint32_t *out, long words_per_row, int L) {
    const long nq = words_per_row / W;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
        for (int l = 0; l < L; ++l) {
            uint32_t *p = out + (long)l * words_per_row + q * W;
            if (W == 1) *p = (uint32_t)q + l;
            if (W == 2) *reinterpret_cast<uint2 *>(p) = make_uint2(q, l);
            if (W == 4) *reinterpret_cast<uint4 *>(p) = make_uint4(q, l, 2, 3);
        }
    }
}
template <int W>
__global__ void __launch_bounds__(256) k_rows_nt(uint32_t *out, long words_per_row, int L) {
    const long nq = words_per_row / W;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
        for (int l = 0; l < L; ++l) {
            uint32_t *p = out + (long)l * words_per_row + q * W;
            __builtin_nontemporal_store((uint32_t)q + l, p);
        }
    }
}
int main() {
    const long n = 10000000;          // elements per row (2 B each)
    const int L = 32;
    const long wpr = n / 2;
    uint32_t *d;
    hipMalloc(&d, (size_t)L * n * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto f) {
        for (int i = 0; i < 3; ++i) f();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 10;
        printf("%-34s %.3f ms  %.2f TB/s\n", name, ms, (double)L * n * 2 / ms / 1e9);
    };
    for (int g : {768, 1536, 4096, 19532}) {
        printf("grid %d\n", g);
        run("linear 16 B / lane", [&] { hipLaunchKernelGGL(k_linear, dim3(g), dim3(256), 0, 0, (uint4 *)d, (long)L * n * 2 / 16); });
        run("rows, 4 B / lane (256 B / wave)", [&] { hipLaunchKernelGGL(k_rows<1>, dim3(g), dim3(256), 0, 0, d, wpr, L); });
        run("rows, 8 B / lane (512 B / wave)", [&] { hipLaunchKernelGGL(k_rows<2>, dim3(g), dim3(256), 0, 0, d, wpr, L); });
        run("rows, 16 B / lane (1 KB / wave)", [&] { hipLaunchKernelGGL(k_rows<4>, dim3(g), dim3(256), 0, 0, d, wpr, L); });
        run("rows, 4 B / lane, nontemporal", [&] { hipLaunchKernelGGL(k_rows_nt<1>, dim3(g), dim3(256), 0, 0, d, wpr, L); });
    }
    return 0;
}

// Calibration of the SQ VALU counters on gfx950: a kernel of a known number of independent 32-bit VALU
// instructions per lane at full occupancy.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_calib.hip -o /tmp/valu_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int WHICH>
__global__ __launch_bounds__(256) void valu_kernel(uint32_t *out, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 ^ 0x1234, a5 = a0 + 99, a6 = a0 * 11, a7 = a0 * 13;
    uint64_t b0 = a0 * 0x9E3779B97F4A7C15ull, b1 = b0 + 7, b2 = b0 ^ 0x55, b3 = b0 * 3;
    for (int i = 0; i < iters; i++) {
        if (WHICH == 0) { // 8 independent 32-bit xor/add chains: 16 VALU per iteration
            a0 = (a0 ^ a1) + i; a1 = (a1 ^ a2) + i; a2 = (a2 ^ a3) + i; a3 = (a3 ^ a4) + i;
            a4 = (a4 ^ a5) + i; a5 = (a5 ^ a6) + i; a6 = (a6 ^ a7) + i; a7 = (a7 ^ a0) + i;
        } else {          // 64-bit compares + selects on 4 chains
            b0 = b0 < b1 ? b0 + i : b1 ^ b2; b1 = b1 < b2 ? b1 + i : b2 ^ b3;
            b2 = b2 < b3 ? b2 + i : b3 ^ b0; b3 = b3 < b0 ? b3 + i : b0 ^ b1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(b0 ^ b1 ^ b2 ^ b3);
}

int main()
{
    uint32_t *out;
    const int blocks = 256 * 8 * 4, iters = 4096;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            if (which == 0) hipLaunchKernelGGL(valu_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(valu_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("which=%d rep=%d ms=%.3f waves=%d iters=%d\n", which, rep, ms, blocks * 4, iters);
        }
    }
    return 0;
}

// A synthetic stand-in for the window kernel's pressure on a CU (round 6): resident workgroups of 512 threads that issue rolling-like
// integer work -- v_alignbit / v_xor / v_add chains, ILP independent chains per thread, optionally one 8-byte LDS read of a 128-byte
// table per step -- for `iters` iterations, on a stream of their own.  tools/hog_experiment.py runs the lookup kernel beside it:
// what slows emit_list_kernel beside the window stage -- the wavefront slots the window kernel holds, or the VALU / LDS cycles it
// takes?  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/valu_hog.hip -o ntlink_amd/build/libvalu_hog.so
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int ILP, bool LDS>
__global__ __launch_bounds__(512) void hog_kernel(uint32_t *out, int iters)
{
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint2 s_tab[16];
    if (threadIdx.x < 16) s_tab[threadIdx.x] = make_uint2(threadIdx.x * 2654435761u, threadIdx.x * 40503u + 77u);
    if (threadIdx.x == 0) s_dyn[0] = 1u;
    __syncthreads();
    uint32_t fx[ILP], ry[ILP];
    for (int c = 0; c < ILP; c++) { fx[c] = threadIdx.x * 2246822519u + c * 97u + blockIdx.x; ry[c] = fx[c] ^ 0x9e3779b9u; }
    uint32_t hits = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < ILP; c++) {
                uint2 sd = make_uint2(0x1234567u + u, 0x7654321u + c);
                if (LDS) sd = s_tab[(fx[c] >> (u + 3)) & 15u];
                const uint32_t fd = fx[c] + fx[c];
                fx[c] = __builtin_amdgcn_alignbit(fx[c], fd, 31) ^ sd.x;
                const uint32_t a = ry[c] ^ sd.y;
                ry[c] = __builtin_amdgcn_alignbit(a >> 1, a, 1);
                const uint32_t key = fx[c] + fx[c] + ry[c];
                hits += key < 0x0A3D70A3u ? 1u : 0u;
            }
        }
    }
    uint32_t acc = hits;
    for (int c = 0; c < ILP; c++) acc ^= fx[c] ^ ry[c];
    if (acc == 0x12345u) out[blockIdx.x] = acc; /* (keeps the work alive) */
}

/* the same work as 256 DIFFERENT unrolled steps per iteration: about 40 KB of straight-line code that every wavefront walks through
   (the window kernel's 64 unrolled rolling steps + lists + scans are 20 KB) -- instruction-cache pressure; GLD: also two 16-byte
   global loads per 64 steps at lane-consecutive addresses, like the strips' base words */
template <bool GLD>
__global__ __launch_bounds__(512) void hog_big_kernel(uint32_t *out, int iters, const uint4 *src, uint32_t nsrc)
{
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint2 s_tab[16];
    if (threadIdx.x < 16) s_tab[threadIdx.x] = make_uint2(threadIdx.x * 2654435761u, threadIdx.x * 40503u + 77u);
    if (threadIdx.x == 0) s_dyn[0] = 1u;
    __syncthreads();
    uint32_t fx = threadIdx.x * 2246822519u + blockIdx.x, ry = fx ^ 0x9e3779b9u, hits = 0;
    uint32_t at = (blockIdx.x * 512u + threadIdx.x) % nsrc;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 256; u++) {
            if (GLD && (u & 63) == 0) {
                const uint4 v = src[at];
                at = (at + 512u * 1024u) % nsrc;
                fx ^= v.x ^ v.w; ry ^= v.y ^ v.z;
            }
            const uint2 sd = s_tab[(fx >> ((u & 7) + 3)) & 15u];
            const uint32_t fd = fx + fx;
            fx = __builtin_amdgcn_alignbit(fx, fd, 31) ^ sd.x ^ (uint32_t)(u * 2654435761u);
            const uint32_t a = ry ^ sd.y;
            ry = __builtin_amdgcn_alignbit(a >> 1, a, 1);
            const uint32_t key = fx + fx + ry;
            hits += key < 0x0A3D70A3u + (uint32_t)u ? 1u : 0u;
        }
    }
    if ((hits ^ fx ^ ry) == 0x12345u) out[blockIdx.x] = hits;
}

/* LDS-heavy: per step three 8-byte LDS reads at lane-random offsets of a 4-KB area (bank conflicts, like the scans' and the lookup
   tables' reads) and one sparse 4-byte write (two lanes in 64, like the candidate push), little VALU */
__global__ __launch_bounds__(512) void hog_lds_kernel(uint32_t *out, int iters)
{
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint2 s_area[512 * 2];
    for (int i = threadIdx.x; i < 1024; i += 512) s_area[i] = make_uint2(i * 2654435761u, i * 40503u + 77u);
    if (threadIdx.x == 0) s_dyn[0] = 1u;
    __syncthreads();
    uint32_t x = threadIdx.x * 2246822519u + blockIdx.x, acc = 0;
    uint32_t *const mine = (uint32_t *)&s_area[512 + (threadIdx.x & ~63)];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint2 a = s_area[x & 511u], b = s_area[(x >> 9) & 511u], c = s_area[(x >> 18) & 511u];
            x = (x * 1664525u + 1013904223u) ^ a.x ^ b.y ^ c.x;
            acc += a.y ^ c.y;
            if (((x >> 5) & 31u) == (threadIdx.x & 31u)) mine[threadIdx.x & 63] = x; /* two lanes of a wavefront */
        }
    }
    if ((acc ^ x) == 0x12345u) out[blockIdx.x] = acc;
}

/* memory-heavy: every wavefront streams 16 bytes per lane per step out of a 1-GB array (coalesced) or gathers them from random lines */
template <bool RANDOM>
__global__ __launch_bounds__(512) void hog_mem_kernel(uint32_t *out, int iters, const uint4 *src, uint32_t nsrc)
{
    uint32_t at = (blockIdx.x * 512u + threadIdx.x) % nsrc, acc = 0, x = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
        const uint4 v = src[at];
        acc ^= v.x ^ v.w;
        x = x * 1664525u + 1013904223u + v.y;
        at = RANDOM ? (x % nsrc) : (at + 512u * 2048u) % nsrc;
    }
    if (acc == 0x12345u) out[blockIdx.x] = acc;
}

static hipStream_t g_stream;
static hipEvent_t g_a, g_b;
static uint32_t *g_out;
static uint4 *g_src;
static uint32_t g_nsrc = 64u << 20; /* 1 GB of uint4 */

extern "C" int hog_start(int wgs, int iters, int ilp, int lds_read, int lds_bytes)
{
    if (!g_stream) {
        if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -1;
        hipEventCreate(&g_a); hipEventCreate(&g_b);
        hipMalloc(&g_out, 1 << 20);
        hipMalloc(&g_src, (size_t)g_nsrc * 16);
        hipMemset(g_src, 1, (size_t)g_nsrc * 16);
    }
    hipEventRecord(g_a, g_stream);
    const dim3 grid(wgs), block(512);
#define GO(I, L) hipLaunchKernelGGL((hog_kernel<I, L>), grid, block, (size_t)lds_bytes, g_stream, g_out, iters)
    if (ilp == 200) hipLaunchKernelGGL(hog_lds_kernel, grid, block, (size_t)lds_bytes, g_stream, g_out, iters);
    else if (ilp == 300) hipLaunchKernelGGL((hog_mem_kernel<false>), grid, block, 0, g_stream, g_out, iters, (const uint4 *)g_src, g_nsrc);
    else if (ilp == 301) hipLaunchKernelGGL((hog_mem_kernel<true>), grid, block, 0, g_stream, g_out, iters, (const uint4 *)g_src, g_nsrc);
    else if (ilp == 100) hipLaunchKernelGGL((hog_big_kernel<false>), grid, block, (size_t)lds_bytes, g_stream, g_out, iters, (const uint4 *)g_src, g_nsrc);
    else if (ilp == 101) hipLaunchKernelGGL((hog_big_kernel<true>), grid, block, (size_t)lds_bytes, g_stream, g_out, iters, (const uint4 *)g_src, g_nsrc);
    else if (lds_read) { if (ilp >= 4) GO(4, true); else if (ilp == 2) GO(2, true); else GO(1, true); }
    else { if (ilp >= 4) GO(4, false); else if (ilp == 2) GO(2, false); else GO(1, false); }
    hipEventRecord(g_b, g_stream);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" float hog_wait(void)
{
    float ms = -1.0f;
    if (hipEventSynchronize(g_b) != hipSuccess) return -1.0f;
    hipEventElapsedTime(&ms, g_a, g_b);
    return ms;
}

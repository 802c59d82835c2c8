/*
 * Mock of the HIP runtime + SIMT vocabulary for running the product's kernels on CPU threads.
 * TEST TOOL ONLY (tests/sim): it lets `pytest -m "not gpu"` execute the exact kernel source
 * (ntlink_amd/csrc) under a pthread-per-lane emulation, and lets CPU sanitizers see it.  It is
 * not a backend: the product loads only the hipcc-built library and fails without a GPU.
 *
 * Model: one workgroup at a time; every HIP thread is a pthread; __syncthreads() is a barrier
 * over the workgroup; wave64 votes/shuffles are barriers over each group of 64 threads;
 * __shared__ variables are function-static storage (one workgroup runs at a time).
 */
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <chrono>
#include <functional>
#include <mutex>
#include <vector>

#define NTL_SIM 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

namespace sim {
struct Block {
    unsigned nthreads;
    pthread_barrier_t bar;
    pthread_barrier_t wbar[16];
    unsigned long long slot[16][64];
};
extern Block *cur;
std::mutex &launch_mutex();
extern unsigned grid_y; /* blockIdx.y of the running launch (2-D grids run one row at a time) */
extern thread_local unsigned tid;
void launch(unsigned grid, unsigned block, const std::function<void()> &fn);
template <typename K, typename... Args>
inline void launch_k(unsigned grid, unsigned block, K kern, Args... args)
{
    launch(grid, block, [=]() { kern(args...); });
}
}  // namespace sim

extern thread_local dim3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

inline void __syncthreads() { pthread_barrier_wait(&sim::cur->bar); }

inline unsigned long long __ballot(int pred)
{
    sim::Block &B = *sim::cur;
    unsigned w = sim::tid >> 6, l = sim::tid & 63;
    unsigned cnt = B.nthreads - 64 * w < 64 ? B.nthreads - 64 * w : 64;
    B.slot[w][l] = pred ? 1 : 0;
    pthread_barrier_wait(&B.wbar[w]);
    unsigned long long m = 0;
    for (unsigned i = 0; i < cnt; i++) m |= (unsigned long long)B.slot[w][i] << i;
    pthread_barrier_wait(&B.wbar[w]);
    return m;
}

template <typename T>
inline T __shfl(T v, int src)
{
    sim::Block &B = *sim::cur;
    unsigned w = sim::tid >> 6, l = sim::tid & 63;
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    B.slot[w][l] = raw;
    pthread_barrier_wait(&B.wbar[w]);
    unsigned long long r = B.slot[w][src & 63];
    pthread_barrier_wait(&B.wbar[w]);
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}

inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }

template <typename T> inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <typename T> inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <typename T> inline T atomicMax(T *p, T v)
{
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
template <typename T> inline T atomicCAS(T *p, T cmp, T val)
{
    __atomic_compare_exchange_n(p, &cmp, val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return cmp;
}

/* ---------------------------------------------------------------- host runtime */
typedef int hipError_t;
#define hipSuccess 0
#define hipErrorNotReady 600
typedef struct sim_stream *hipStream_t;
typedef struct sim_event { std::chrono::steady_clock::time_point t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { char name[64]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[64]; };

inline const char *hipGetErrorString(hipError_t) { return "sim"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 8; return hipSuccess; } /* eight identical mock devices (one per test rank) */
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof(*p)); strcpy(p->name, "simt-emulation"); strcpy(p->gcnArchName, "sim");
    p->multiProcessorCount = 1; p->totalGlobalMem = (size_t)8 << 30; return hipSuccess;
}
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
#define hipHostMallocDefault 0u
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
#define hipStreamNonBlocking 1u
#define hipEventDisableTiming 2u
#define hipEventBlockingSync 1u
/* two distinguishable handles so that the product's two-stream bookkeeping (events, per-stream block caches) is exercised;
   the mock itself runs every launch at once, in program order */
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { static char tag[64]; static int n = 0; *s = (hipStream_t)&tag[n++ & 63]; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new sim_event; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...)                         \
    do {                                                                                  \
        dim3 g_ = (grid), b_ = (block);                                                   \
        std::lock_guard<std::mutex> launch_lock_(sim::launch_mutex()); /* one launch at a time, whatever host thread */ \
        for (unsigned y_ = 0; y_ < g_.y; y_++) {                                          \
            sim::grid_y = y_;                                                             \
            sim::launch_k(g_.x, b_.x, kern, __VA_ARGS__);                                 \
        }                                                                                 \
    } while (0)

#!/usr/bin/env python3
"""What page-locking a staging buffer costs, by how the memory is obtained (the readers of pipeline.run_pair page-lock 1.4 GB on a
process's first pass): hipHostMalloc against hipHostRegister of an anonymous mapping, with and without transparent huge pages,
pre-faulted or not; and the first / second DMA out of each.  usage: tools/pin_bench.py [MB]"""
import ctypes as C
import mmap
import sys
import time

hip = C.CDLL("libamdhip64.so")
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 146
n = MB << 20
libc = C.CDLL("libc.so.6", use_errno=True)
libc.mmap.restype = C.c_void_p
libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
libc.memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
assert hip.hipSetDevice(0) == 0
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), n) == 0
hip.hipDeviceSynchronize()
print("transparent_hugepage:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())


def lap(t0):
    return round((time.perf_counter() - t0) * 1e3, 1)


def dma(p):
    t0 = time.perf_counter()
    assert hip.hipMemcpy(d, p, n, 1) == 0
    a = lap(t0)
    t0 = time.perf_counter()
    assert hip.hipMemcpy(d, p, n, 1) == 0
    return a, lap(t0)


for rep in range(2):
    p = C.c_void_p()
    t0 = time.perf_counter()
    assert hip.hipHostMalloc(C.byref(p), n, 0) == 0
    t_alloc = lap(t0)
    t0 = time.perf_counter()
    libc.memset(p, 1, n)
    t_touch = lap(t0)
    print(f"hipHostMalloc {MB} MB: alloc {t_alloc} ms, first touch {t_touch} ms, DMA first/second {dma(p)} ms")
    hip.hipHostFree(p)
    for thp, prefault in ((0, 0), (0, 1), (1, 0), (1, 1)):
        t0 = time.perf_counter()
        size = n + (2 << 20)
        base = libc.mmap(None, size, 3, 0x22, -1, 0)  # PROT_READ|WRITE, MAP_PRIVATE|MAP_ANONYMOUS
        q = (base + (2 << 20) - 1) & ~((2 << 20) - 1)
        if thp:
            libc.madvise(q, n, 14)  # MADV_HUGEPAGE
        if prefault:
            libc.memset(q, 1, n)
        t_map = lap(t0)
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(q, n, 0)
        t_reg = lap(t0)
        t0 = time.perf_counter()
        libc.memset(q, 2, n)
        t_touch = lap(t0)
        print(f"mmap thp={thp} prefault={prefault}: map {t_map} ms, hipHostRegister {t_reg} ms (rc {rc}), touch {t_touch} ms, DMA first/second {dma(q) if rc == 0 else None} ms")
        if rc == 0:
            hip.hipHostUnregister(q)
        libc.munmap(base, size)

#!/usr/bin/env python3
"""Where the time of 16 .fq.gz files goes on the GPU box's host (no device work): raw libdeflate per file on 1 / 16 threads, the
native reader alone with 2 / 16 / 32 files opened ahead.  usage: tools/gz_diag.py [n_files] [Mbases per file]"""
import ctypes
import glob
import os
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 125
d = tempfile.mkdtemp(prefix="ntl_gzd_", dir="/dev/shm")
acgt = np.frombuffer(b"ACGT", np.uint8)


def make(i):
    rng = np.random.default_rng(i)
    n, L = mb * 1_000_000, 15000
    seq = acgt[rng.integers(0, 4, n)].tobytes()
    data = b"".join(b"@r%d_%d\n" % (i, j) + seq[j * L:(j + 1) * L] + b"\n+\n" + b"I" * L + b"\n" for j in range(n // L))
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    z = co.compress(data) + co.flush()
    open(os.path.join(d, f"r{i:02d}.fq.gz"), "wb").write(z)
    return len(data), len(z)


try:
    t = time.time()
    with ThreadPoolExecutor(16) as ex:
        sizes = list(ex.map(make, range(nf)))
    print("made", nf, "files", sizes[0], round(time.time() - t, 1), "s", flush=True)
    files = sorted(glob.glob(os.path.join(d, "*.fq.gz")))
    L = ctypes.CDLL("libdeflate.so.0")
    L.libdeflate_alloc_decompressor.restype = ctypes.c_void_p
    L.libdeflate_gzip_decompress_ex.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                                ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    zs = [open(f, "rb").read() for f in files]
    outs = [ctypes.create_string_buffer(sizes[0][0] + 1024) for _ in files]

    def inflate(i):
        dd = L.libdeflate_alloc_decompressor()
        a, b = ctypes.c_size_t(0), ctypes.c_size_t(0)
        t0 = time.time()
        L.libdeflate_gzip_decompress_ex(dd, zs[i], len(zs[i]), outs[i], len(outs[i]), ctypes.byref(a), ctypes.byref(b))
        return time.time() - t0

    print("raw inflate, one file alone: %.3f s" % inflate(0), flush=True)
    for T in (4, 8, 16, 32):
        t = time.time()
        with ThreadPoolExecutor(T) as ex:
            per = list(ex.map(inflate, range(nf)))
        dt = time.time() - t
        print(f"raw inflate, {nf} files on {T} threads: {dt:.3f} s wall, per file {min(per):.3f}-{max(per):.3f} s, {sum(s[0] for s in sizes) / dt / 1e9:.2f} GB/s of text", flush=True)
    from ntlink_amd import seqio
    for ahead in (2, 16, 32):
        for rep in range(2):
            t = time.time()
            nb, st = 0, {}
            for rs in seqio.load(files, max_bases=512_000_000, packed=True, ahead=ahead, stats=st):
                nb += rs.bases
            dt = time.time() - t
            print("reader alone, ahead", ahead, "s", round(dt, 3), "Gbases/s", round(nb / dt / 1e9, 3),
                  {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}, flush=True)
    for readers in (2,):
        t = time.time()
        nb = sum(rs.bases for rs in seqio.load_parallel(files, readers=readers, max_bases=512_000_000, packed=True))
        dt = time.time() - t
        print("load_parallel readers", readers, "s", round(dt, 3), "Gbases/s", round(nb / dt / 1e9, 3), flush=True)
finally:
    import shutil
    shutil.rmtree(d, ignore_errors=True)

#!/usr/bin/env python3
"""Kill-criterion measurement of the device inflate experiment: FASTQ-like text, bgzip'd, every member inflated by
one lane (ntl_bgzf_inflate); text GB/s of the kernel alone against zlib on the host's cores.
usage: tools/inflate_bench.py [--mbytes 1024] [--level 6]"""
import argparse
import json
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ntlink_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mbytes", type=int, default=1024)
ap.add_argument("--level", type=int, default=6)
a = ap.parse_args()
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", np.uint8)


def fastq(seed, nbytes):
    r = np.random.default_rng(seed)
    out, n, i = [], 0, 0
    while n < nbytes:
        ln = int(r.integers(5000, 30000))
        rec = b"@read_%d_%d runid=0123456789abcdef ch=%d start_time=2026-01-01T00:00:00Z\n" % (seed, i, i % 512) + bytes(acgt[r.integers(0, 4, ln)]) + b"\n+\n" + \
            bytes((np.clip(r.normal(20, 6, ln), 2, 40).astype(np.uint8) + 33)) + b"\n"
        out.append(rec)
        n += len(rec)
        i += 1
    return b"".join(out)


import struct  # noqa: E402


def member(ch, level):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(ch) + co.flush()
    return b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) + body + \
        struct.pack("<II", zlib.crc32(ch) & 0xFFFFFFFF, len(ch))


t0 = time.time()
with ThreadPoolExecutor(16) as ex:
    parts = list(ex.map(lambda s: fastq(s, 16 << 20), range(max(1, a.mbytes // 16))))
text = b"".join(parts)
chunks = [text[i:i + 0xFF00] for i in range(0, len(text), 0xFF00)]
with ThreadPoolExecutor(16) as ex:
    comp = b"".join(ex.map(lambda c: member(c, a.level), chunks)) + member(b"", a.level)
prep = time.time() - t0
dev = capi.Device(0)
res = []
for rep in range(3):
    t0 = time.time()
    got, ms, n, bad = dev.bgzf_inflate(comp)
    wall = time.time() - t0
    res.append({"kernel_ms": round(ms, 3), "text_GB_per_s": round(len(text) / ms / 1e6, 2), "call_s_incl_pcie_both_ways": round(wall, 3)})
ok = bad == 0 and bytes(got[:1 << 20]) == text[:1 << 20] and bytes(got[-(1 << 20):]) == text[-(1 << 20):] and len(got) == len(text)
# the host: zlib on 16 threads (libdeflate is about twice as fast per core)
members = []
at = 0
while at < len(comp):
    bs = struct.unpack_from("<H", comp, at + 16)[0] + 1
    members.append(comp[at + 18:at + bs - 8])
    at += bs
t0 = time.time()
with ThreadPoolExecutor(16) as ex:
    n_out = sum(ex.map(lambda m: len(zlib.decompress(m, -15)) if m else 0, members))
host = time.time() - t0
print(json.dumps({"text_bytes": len(text), "compressed_bytes": len(comp), "members": n, "zlib_level": a.level, "identical": bool(ok), "failed_members": bad,
                  "device": res, "host_zlib_16_threads_GB_per_s": round(n_out / host / 1e9, 2), "prepare_s": round(prep, 1),
                  "kill_criterion": "slower than the host's libdeflate on 16 cores = 13.6 GB/s of text (profiles/r03ah_gz_diag.txt)"}))
dev.close()

#!/usr/bin/env python3
"""Host-side ingest diagnostics (no GPU work): where the time of the FASTA reader goes on this machine."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ntlink_amd import capi, seqio, synth  # noqa: E402
from e2e_bench import write_fasta  # noqa: E402


def main():
    out = {"cores": os.cpu_count()}
    for f in ("enabled", "defrag", "shmem_enabled"):
        try:
            out["thp_" + f] = open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip()
        except OSError as e:
            out["thp_" + f] = str(e)
    W = synth.workload("C2", 1.0)
    chroms, cbuf, coff, cn, _ = synth.make_assembly(1, W["n_chrom"], W["contigs_per_chrom"], W["contig_len"])
    rbuf, roff, rn = synth.make_reads(2, chroms, W["read_bases"], W["read_len"], W["sub"], W["ins"], W["dele"], lognormal_sigma=0.4)
    path = "/tmp/ntl_diag_reads.fa"
    write_fasta(path, rbuf, roff, rn)
    nbytes = os.path.getsize(path)
    out["file_bytes"] = nbytes
    # raw memory numbers
    a = np.empty(nbytes, np.uint8); t = time.perf_counter(); a[:] = 1; out["touch_np_empty_s"] = round(time.perf_counter() - t, 4)
    t = time.perf_counter(); a[:] = 2; out["retouch_s"] = round(time.perf_counter() - t, 4)
    b = np.empty(nbytes, np.uint8); b[:] = 1
    t = time.perf_counter(); b[:] = a; out["memcpy_1thr_s"] = round(time.perf_counter() - t, 4)
    del a, b
    L = capi.load()
    rows = []
    for thr in ("1", "4", "8", "16", "32", "64", "128"):
        os.environ["NTL_IO_THREADS"] = thr
        for dest in ("np", "reuse"):
            best = None
            keep = None
            for rep in range(3):
                h = C.c_void_p(); L.ntl_fastx_open(path.encode(), C.byref(h))
                n = C.c_uint64(); t0 = time.perf_counter(); L.ntl_fastx_next(h, 0, C.byref(n)); t1 = time.perf_counter()
                nb, nn = C.c_uint64(), C.c_uint64()
                L.ntl_fastx_sizes(h, None, C.byref(nb), C.byref(nn))
                if dest == "reuse" and keep is not None:
                    buf = keep
                else:
                    buf = np.empty(nb.value, np.uint8)
                keep = buf
                names = np.empty(nn.value, np.uint8)
                off, noff = np.empty(n.value + 1, np.uint64), np.empty(n.value + 1, np.uint64)
                t2 = time.perf_counter()
                L.ntl_fastx_copy(h, buf.ctypes.data, off.ctypes.data, names.ctypes.data, noff.ctypes.data)
                t3 = time.perf_counter()
                L.ntl_fastx_close(h)
                cur = (round(t1 - t0, 4), round(t3 - t2, 4))
                if best is None or sum(cur) < sum(best):
                    best = cur
                if dest != "reuse":
                    del buf
            rows.append({"threads": thr, "dest": dest, "count_pass_s": best[0], "write_pass_s": best[1]})
    out["reader"] = rows
    print(json.dumps(out))


if __name__ == "__main__":
    main()

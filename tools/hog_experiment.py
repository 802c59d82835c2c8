#!/usr/bin/env python3
"""What slows the lookup kernel beside the window stage (round 6)?  emit_list_kernel on one C3 sub-batch (3.9 Gbases, the full index), on one
stream, beside a synthetic resident kernel (tools/valu_hog.hip) that holds N wavefronts per CU and issues rolling-like integer work with 1 / 2 / 4
independent chains per thread, with / without the window kernel's LDS seed reads and LDS footprint.  One JSON line per setting:
the lookup kernel's span (ms per launch) and the hog's time.  usage: tools/hog_experiment.py [--reps 3]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NTL_PIPELINE"] = "0"  # the sketch's own kernels one after the other on ONE stream: only the hog runs beside them
from ntlink_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = capi.Device(0)
hog = C.CDLL(os.path.join(ROOT, "ntlink_amd", "build", "libvalu_hog.so"))
hog.hog_wait.restype = C.c_float
W = synth.workload("C3", 1.0)
wl = synth.DeviceWorkload(dev, "C3", 1.0, read_bases=3_950_000_000, batch_bases=3_950_000_000, read_seed=2)
k, w = W["k"], W["w"]
csk = dev.sketch(wl.contigs, k, w)
ix = dev.index(csk, wl.ctg_len)
rb = wl.read_batches[0]
dev.sketch(rb, k, w, index=ix, records=False).close()
dev.sync()
n_cu = 256
for name, (wgs_per_cu, ilp, lds_read, lds_bytes) in [("none", (0, 1, 0, 0)),
                                                     ("LDS-heavy, 8 waves", (1, 200, 1, 20000)), ("LDS-heavy, 16 waves", (2, 200, 1, 20000)), ("LDS-heavy, 24 waves", (3, 200, 1, 20000)),
                                                     ("streaming global reads, 8 waves", (1, 300, 0, 0)), ("streaming global reads, 24 waves", (3, 300, 0, 0)),
                                                     ("random 16-byte global reads, 8 waves", (1, 301, 0, 0)), ("random 16-byte global reads, 24 waves", (3, 301, 0, 0)),
                                                     ("24 waves, 40 KB of code", (3, 100, 1, 30000)), ("24 waves, 40 KB of code + global loads", (3, 101, 1, 30000)),
                                                     ("16 waves, 40 KB of code + global loads", (2, 101, 1, 30000)),
                                                     ("16 waves ilp1", (2, 1, 0, 1024)), ("16 waves ilp2", (2, 2, 0, 1024)), ("16 waves ilp4", (2, 4, 0, 1024)),
                                                     ("16 waves ilp1 +lds reads, 31 KB", (2, 1, 1, 30000)), ("16 waves ilp2 +lds reads, 31 KB", (2, 2, 1, 30000)),
                                                     ("16 waves ilp4 +lds reads, 31 KB", (2, 4, 1, 30000)),
                                                     ("24 waves ilp1 +lds reads, 31 KB", (3, 1, 1, 30000)), ("24 waves ilp2 +lds reads, 31 KB", (3, 2, 1, 30000)),
                                                     ("24 waves ilp4 +lds reads, 31 KB", (3, 4, 1, 30000)), ("32 waves ilp1 +lds reads, 31 KB", (4, 1, 1, 30000))]:
    dev.prof_enable(True); dev.prof_reset()
    hog_ms = None
    t0 = time.perf_counter()
    for _ in range(a.reps):
        if wgs_per_cu:
            iters = int(60000 / ilp) if ilp < 100 else (2000 if ilp < 200 else (40000 if ilp == 200 else 30000))
            assert hog.hog_start(wgs_per_cu * n_cu, iters, ilp, lds_read, lds_bytes) == 0
            time.sleep(0.002)
        sk = dev.sketch(rb, k, w, index=ix, records=False)
        sk.wait(); sk.close()
        dev.sync()
        if wgs_per_cu:
            hog_ms = float(hog.hog_wait())
    out = {nm: round(dev.prof_get(nm)[0] / a.reps, 3) for nm in ("sketch_wave", "sketch_mask", "sketch_emit")}
    dev.prof_enable(False)
    print(json.dumps({"beside": name, "ms_per_launch": out, "hog_ms": hog_ms,
                      "hog_GBps": round(wgs_per_cu * n_cu * 512 * 16 * 30000 / (hog_ms * 1e-3) / 1e9) if hog_ms and ilp >= 300 else None}), flush=True)
dev.close()

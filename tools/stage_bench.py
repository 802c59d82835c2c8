#!/usr/bin/env python3
"""Stage times of ONE read batch of the pair driver on the device (page-locked ASCII bases -> records on the host), one
stream and two contexts side by side: where the driver's ~11 ms per 256-Mbase batch go."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ntlink_amd import capi, synth  # noqa: E402

dev = capi.Device(0)
wl = synth.DeviceWorkload(dev, "C3", with_reads=False)
W = wl.W
k, w = W["k"], W["w"]
csk = dev.sketch(wl.contigs, k, w)
ix = dev.index(csk, wl.ctg_len)
rb, rlen = wl.make_reads(256_000_000, seed=(5, 1))
buf, off = rb.download()
rb.close()
pin = dev.pinned_empty(len(buf))
pin[:] = buf
out = {"bases": int(off[-1]), "reads": len(rlen)}


def one(d, src, T):
    t0 = time.perf_counter(); b = d.batch(src, off); t1 = time.perf_counter()
    sk = d.sketch(b, k, w); t2 = time.perf_counter()
    res = d.map(ix, sk, rlen, k=k); t3 = time.perf_counter()
    rec = res.download(pinned=True); t4 = time.perf_counter()
    d.pinned_release(rec["_pinned"])
    for h in (res, sk, b):
        h.close()
    for name, v in (("upload_pack", t1 - t0), ("sketch", t2 - t1), ("map", t3 - t2), ("download", t4 - t3), ("total", t4 - t0)):
        T.setdefault(name, []).append(v)


for label, src in (("pinned", pin), ("pageable", buf)):
    T = {}
    for _ in range(6):
        one(dev, src, T)
    out[label + "_ms"] = {n: round(1000 * float(np.median(v[1:])), 3) for n, v in T.items()}
# raw copy rate of the same bytes
import ctypes as C
t0 = time.perf_counter()
for _ in range(5):
    b = dev.batch(pin, off); b.close()
out["batch_create_only_ms"] = round(1000 * (time.perf_counter() - t0) / 5, 3)
dev.prof_enable(True); dev.prof_reset()
b = dev.batch(pin, off); b.close()
out["batch_pack_device_ms"] = round(dev.prof_get("batch_pack")[0], 3)
dev.prof_enable(False)
# two contexts, two threads
d2 = dev.clone()
pin2 = d2.pinned_empty(len(buf)); pin2[:] = buf
for n_thr in (1, 2):
    T1, T2 = {}, {}
    t0 = time.perf_counter()
    ths = [threading.Thread(target=lambda d=d, p=p, T=T: [one(d, p, T) for _ in range(8)]) for d, p, T in ((dev, pin, T1), (d2, pin2, T2))[:n_thr]]
    for t in ths: t.start()
    for t in ths: t.join()
    out[f"{n_thr}_threads_ms_per_batch"] = round(1000 * (time.perf_counter() - t0) / (8 * n_thr), 3)
    out[f"{n_thr}_threads_stage_ms"] = {n: round(1000 * float(np.median(v[1:])), 3) for n, v in T1.items()}
print(json.dumps(out))
for h in (ix, csk):
    h.close()
wl.close()
d2.close()
dev.close()

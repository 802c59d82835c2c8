#!/usr/bin/env python3
"""Volume soak of the MAPPING half on the GPU (the sketch has tests/gpu_volume_soak.py): read batches of a BASELINE workload
generated on the device (other seeds than the tests and the bench) are sketched for the index and mapped the way the pair
driver and the bench do it -- sketch and map of batch i+1 are queued BEFORE the records of batch i are asked for, so the two
streams of a context really overlap -- and every mapping, hit and PAF record is compared with the oracle's (its own sketch of
the downloaded bases, its own index of the whole assembly, its own map loop).  C5 (HiFi, --sensitive, h = 0.92: all three LDS
size classes of the map kernels and the global-scratch path) is the default.
records=0: the read sketches are made with Device.sketch(..., index=ix, records=False) -- ntl_sketch_run_for_map, the call form of
the pair driver and of bench.py: no minimizer records exist on the device, the mappings alone are compared.
Usage: tests/gpu_map_soak.py [workload=C5] [batches=10] [bases per batch=1e9] [seed0=300] [records=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import parity_cases as pc  # noqa: E402
from ntlink_amd import capi, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bases = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1_000_000_000
seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 300
records = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
dev = capi.Device(0)
wl = synth.DeviceWorkload(dev, name, with_reads=False)
W = wl.W
k, w = W["k"], W["w"]
kw = dict(k=k, z=1000, x=0.0, sensitive=W["sensitive"], repeat_filter=False)
t0 = time.time()
csk = dev.sketch(wl.contigs, k, w)
ix = dev.index(csk, wl.ctg_len)
cbuf, coff = wl.contigs.download()
co, ch, cp, cs = oracle.sketch_batch(cbuf, coff, k, w, threads=cores)
d_off, d_h, d_p, d_s = csk.download()
assert np.array_equal(co, d_off) and np.array_equal(ch, d_h) and np.array_equal(cp, d_p) and np.array_equal(cs, d_s), "contig sketch differs"
oix = oracle.Index(ch, pc.contig_ids(co), cp, cs)
assert len(ix) == len(oix), "index size differs"
del cbuf
print(f"{name}: {len(wl.ctg_len)} contigs, {len(ch)} contig minimizers, index {len(oix)} keys, pipelined={dev.pipelined}, {time.time() - t0:.0f} s", flush=True)

tot = nmaps = nhits = npafs = 0
classes = np.zeros(4, np.int64)
prev = None


def check(item):
    global tot, nmaps, nhits, npafs
    b, rbuf, roff, rlen, rsk, res = item
    got = res.download()
    assert rsk.has_records == records
    if records:
        off, h, p, s = rsk.download()
    res.close(); rsk.close()
    qoff, qh, qp, qs = oracle.sketch_batch(rbuf, roff, k, w, threads=cores)
    if records and not (np.array_equal(off, qoff) and np.array_equal(h, qh) and np.array_equal(p, qp) and np.array_equal(s, qs)):
        print(f"SKETCH MISMATCH workload {name} seed ({seed0}, {b})")
        sys.exit(1)
    exp = oracle.map_reads(oix, wl.ctg_len, qoff, rlen, qh, qp, qs, threads=cores, **kw)
    try:
        pc.assert_same_records(got, exp)
    except AssertionError as exc:
        print(f"MAP MISMATCH workload {name} seed ({seed0}, {b}): {exc}")
        sys.exit(1)
    nmx = np.diff(qoff.astype(np.int64))
    classes[:] += [int((nmx <= 256).sum()), int(((nmx > 256) & (nmx <= 512)).sum()), int(((nmx > 512) & (nmx <= 1024)).sum()), int((nmx > 1024).sum())]
    tot += int(roff[-1]); nmaps += len(got["maps"]); nhits += len(got["hits"]); npafs += len(got["pafs"])
    print(f"batch {b}: {int(roff[-1])} bases, {len(got['maps'])} mappings / {len(got['hits'])} hits / {len(got['pafs'])} PAF records equal, {time.time() - t0:.0f} s", flush=True)


for b in range(batches):
    rb, rlen = wl.make_reads(bases, seed=(seed0, b))
    rsk = dev.sketch(rb, k, w, index=ix, records=records)  # queued
    res = dev.map(ix, rsk, rlen, **kw)         # queued behind it: nothing has waited yet
    rbuf, roff = rb.download()
    rb.close()
    if prev is not None:
        check(prev)                            # the previous batch's records are asked for with this batch's kernels in flight
    prev = (b, rbuf, roff, rlen, rsk, res)
check(prev)
dev.sync()
print(f"map soak clean: {name} k{k} w{w} sensitive={W['sensitive']} records={records}, {tot} bases, {nmaps} mappings, {nhits} hits, {npafs} PAF records; "
      f"reads by minimizer count <=256 / <=512 / <=1024 / more: {classes.tolist()}; {time.time() - t0:.0f} s")

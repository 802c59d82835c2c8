#!/usr/bin/env python3
"""Volume soak of the sketch on the GPU: the rare paths of the 32-bit window pass (keys within 3 of a window minimum: about one
window in 2^29) only show up in gigabases.  Generates read batches of the C3 / C5 workloads on the device (other seeds than the
tests and the bench), sketches them on the device and with the oracle, compares every record, and reports how many strips took
the exact pass.  Usage: tests/gpu_volume_soak.py [workload=C3] [batches=8] [bases per batch=1.5e9] [seed0=100] [k] [w]
(k, w: other sketch parameters than the workload's, on the same reads)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from ntlink_amd import capi, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bases = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1_500_000_000
seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = capi.Device(0)
wl = synth.DeviceWorkload(dev, name, with_reads=False)
k, w = wl.W["k"], wl.W["w"]
if len(sys.argv) > 6:
    k, w = int(sys.argv[5]), int(sys.argv[6])
t0 = time.time()
tot = redo = mx = 0
for b in range(batches):
    rb, rlen = wl.make_reads(bases, seed=(seed0, b))
    rbuf, roff = rb.download()
    with dev.sketch(rb, k, w) as sk:
        off, h, p, s = sk.download()
        redo += sk.redo_strips
    rb.close()
    ooff, oh, op, os_ = oracle.sketch_batch(rbuf, roff, k, w)
    if not (np.array_equal(off, ooff) and np.array_equal(h, oh) and np.array_equal(p, op) and np.array_equal(s, os_)):
        print(f"SKETCH MISMATCH workload {name} seed ({seed0}, {b}): {len(h)} records, oracle {len(oh)}")
        # the reads whose lists differ, for a replay under the SIMT mock (NTL_SOAK_DUMP=<file>)
        cnt, ocnt = np.diff(off.astype(np.int64)), np.diff(ooff.astype(np.int64))
        badr = [r for r in range(len(cnt)) if cnt[r] != ocnt[r] or not np.array_equal(p[off[r]:off[r + 1]], op[ooff[r]:ooff[r + 1]])
                or not np.array_equal(h[off[r]:off[r + 1]], oh[ooff[r]:ooff[r + 1]])]
        print(f"{len(badr)} reads differ; first: {badr[:5]}")
        for r in badr[:3]:
            mine, theirs = p[off[r]:off[r + 1]].tolist(), op[ooff[r]:ooff[r + 1]].tolist()
            print(f"read {r} len {int(roff[r + 1] - roff[r])}: device-only positions {sorted(set(mine) - set(theirs))}, oracle-only {sorted(set(theirs) - set(mine))}")
        if os.environ.get("NTL_SOAK_DUMP"):
            with open(os.environ["NTL_SOAK_DUMP"], "wb") as fh:
                for r in badr[:8]:
                    fh.write(b">r%d\n" % r + bytes(rbuf[roff[r]:roff[r + 1]]) + b"\n")
        sys.exit(1)
    tot += int(roff[-1]); mx += len(h)
    print(f"batch {b}: {int(roff[-1])} bases, {len(h)} minimizers equal, redo strips so far {redo}, {time.time() - t0:.0f} s", flush=True)
print(f"volume soak clean: {name} k{k} w{w}, {tot} bases, {mx} minimizers, {redo} strips through the exact pass, {time.time() - t0:.0f} s")

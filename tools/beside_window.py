#!/usr/bin/env python3
"""The lookup kernel beside the REAL window kernel of another context (round 6; tools/hog_experiment.py uses a synthetic kernel): a thread
keeps sketching one read batch on context B (window kernel, block-minima pass, emit without lookups) while context A, one stream, sketches
another batch for the index; A's `sketch_emit` span = emit_list_kernel<1> beside whatever B runs.  B's launch shape by env per setting.
usage: tools/beside_window.py"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NTL_PIPELINE"] = "0"
from ntlink_amd import capi, synth  # noqa: E402

devA, devB = capi.Device(0), capi.Device(0)
W = synth.workload("C3", 1.0)
wl = synth.DeviceWorkload(devA, "C3", 1.0, read_bases=3_950_000_000, batch_bases=3_950_000_000, read_seed=2)
wlB = synth.DeviceWorkload(devB, "C3", 1.0, read_bases=3_950_000_000, batch_bases=3_950_000_000, read_seed=3)
k, w = W["k"], W["w"]
csk = devA.sketch(wl.contigs, k, w)
ix = devA.index(csk, wl.ctg_len)
rb, rbB = wl.read_batches[0], wlB.read_batches[0]
devA.sketch(rb, k, w, index=ix, records=False).close(); devA.sync()
devB.sketch(rbB, k, w).close(); devB.sync()
stop = False


def loopB():
    while not stop:
        sk = devB.sketch(rbB, k, w)
        sk.wait(); sk.close()


for name, env in [("nothing", None), ("window kernel, 32 wavefronts per CU", {}), ("window kernel, 24 per CU", {"NTL_SKW_WGS_PER_CU": "3"}),
                  ("window kernel, 16 per CU", {"NTL_SKW_WGS_PER_CU": "2"}), ("window kernel, 8 per CU", {"NTL_SKW_WGS_PER_CU": "1"}),
                  ("workgroup-per-strip threshold kernel", {"NTL_SKETCH_WAVE": "0"})]:
    for key in ("NTL_SKW_WGS_PER_CU", "NTL_SKETCH_WAVE"):
        os.environ.pop(key, None)
    th = None
    if env is not None:
        os.environ.update(env)
        stop = False
        th = threading.Thread(target=loopB); th.start()
        time.sleep(0.05)
    devA.prof_enable(True); devA.prof_reset()
    devB.prof_enable(True); devB.prof_reset()
    reps = 10
    for _ in range(reps):
        sk = devA.sketch(rb, k, w, index=ix, records=False)
        sk.wait(); sk.close()
    devA.sync()
    out = {nm: round(devA.prof_get(nm)[0] / reps, 3) for nm in ("sketch_wave", "sketch_emit")}
    if th:
        stop = True; th.join(); devB.sync()
        nB = devB.prof_get("sketch_wave")[1]
        outB = {nm: round(devB.prof_get(nm)[0] / max(devB.prof_get(nm)[1], 1), 3) for nm in ("sketch_wave", "sketch_mask", "sketch_emit")}
    else:
        outB = None
    devA.prof_enable(False); devB.prof_enable(False)
    print(json.dumps({"beside": name, "A_ms_per_launch": out, "B_ms_per_launch": outB}), flush=True)

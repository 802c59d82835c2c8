#!/usr/bin/env python3
"""How the two streams share the device: one workload made once, the bench's step timed under a list of settings of the launch
parameters that are read per call (NTL_SKW_BUDGET, NTL_EMIT_WGS_PER_CU, NTL_SKW_WGS_PER_CU, NTL_BENCH_RECORDS ...).
usage: tools/share_sweep.py --workload C3 --steps 3 'NTL_SKW_BUDGET=3 NTL_EMIT_WGS_PER_CU=4' 'NTL_SKW_BUDGET=0' ...
An empty string is the default configuration.  One JSON line per setting."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ntlink_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch-bases", type=float, default=3.95e9)
ap.add_argument("settings", nargs="*", default=[""])
a = ap.parse_args()
dev = capi.Device(0)
W = synth.workload(a.workload, 1.0)
wl = synth.DeviceWorkload(dev, a.workload, 1.0, read_bases=W["read_bases"], batch_bases=int(a.batch_bases), read_seed=2)
k, w = W["k"], W["w"]
params = dict(k=k, z=1000, x=0.0, sensitive=W["sensitive"], repeat_filter=False)
csk = dev.sketch(wl.contigs, k, w)
ix = dev.index(csk, wl.ctg_len)
dev.sync()
STAGES = ("sketch_meta", "sketch_mask", "sketch_wave", "sketch_redo", "sketch_emit", "probe", "map", "compact")


def step():
    held = []
    for rb, rl in zip(wl.read_batches, wl.read_lens):
        rsk = dev.sketch(rb, k, w, index=ix, records=os.environ.get("NTL_BENCH_RECORDS", "0") == "1")
        res = dev.map(ix, rsk, rl, **params)
        held.append((rsk, res))
        while len(held) > 2:
            s, r = held.pop(0)
            r.close(); s.close()
    for s, r in held:
        r.close(); s.close()
    dev.sync()


base_env = dict(os.environ)
for setting in a.settings:
    os.environ.clear(); os.environ.update(base_env)
    for kv in setting.split():
        key, v = kv.split("=", 1)
        os.environ[key] = v
    step()
    dev.prof_enable(True); dev.prof_reset()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    dt = (time.perf_counter() - t0) / a.steps
    prof = {nm: round(dev.prof_get(nm)[0] / a.steps, 2) for nm in STAGES}
    dev.prof_enable(False)
    print(json.dumps({"workload": a.workload, "setting": setting, "ms_per_step": round(dt * 1e3, 2),
                      "Gbases_per_s": round(wl.read_bases / dt / 1e9, 1), "spans_ms": prof}), flush=True)

#!/usr/bin/env python3
"""Sketch stage alone on one device-generated read batch: per-kernel-group times from HIP events (A/B of the window
pass, PMC passes of one kernel without the rest of the bench).  usage: tools/sketch_bench.py [--bases N] [--k K] [--w W] [--reps R]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ntlink_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bases", type=float, default=3.9e9)
ap.add_argument("--read-len", type=int, default=15000)
ap.add_argument("--k", type=int, default=32)
ap.add_argument("--w", type=int, default=250)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = capi.Device(0)
plan = synth.plan_assembly(1, 4, 50, 600000)
g = dev.synth_genome(1, plan["chrom_len"])
rp = synth.plan_reads(3, plan["chrom_len"], a.bases, a.read_len)
rb = dev.synth_slices(g, 11, rp["chrom"], rp["start"], rp["length"], rp["reverse"], sub=0.02, ins=0.015, dele=0.015)
dev.sketch(rb, a.k, a.w).close()
dev.prof_enable(True)
dev.prof_reset()
for _ in range(a.reps):
    sk = dev.sketch(rb, a.k, a.w)
    n, strips, redo = sk.count, sk.strips, sk.redo_strips
    sk.close()
out = {nm: round(dev.prof_get(nm)[0] / a.reps, 3) for nm in ("sketch_meta", "sketch_mask", "sketch_redo", "sketch_emit")}
bases = int(rp["length"].sum())
print(json.dumps({"bases": bases, "k": a.k, "w": a.w, "minimizers": n, "strips": strips, "redo_strips": redo, "ms": out,
                  "window_pass_Gbases_per_s": round(bases / out["sketch_mask"] / 1e6, 1), "env": {k: v for k, v in os.environ.items() if k.startswith("NTL_")}}))
dev.close()

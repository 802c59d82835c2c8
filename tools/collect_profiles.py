#!/usr/bin/env python3
"""Copies the summaries of one GPU round (gpurun_out/<tag>/, written by tools/gpu_round.sh) into profiles/
and refreshes profiles/traffic.json (HBM bytes per sketch_mask_kernel launch from the PMC passes)."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
out = "profiles"
os.makedirs(out, exist_ok=True)


def pmc(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: {"launches": len(x), "avg": sum(x) / len(x), "max": max(x)} for c, x in v.items()} for k, v in agg.items()}


summary = {}
for t in ("pmc_fetch", "pmc_write", "pmc_sq"):
    p = os.path.join(src, t, "p_counter_collection.csv")
    if os.path.exists(p):
        summary[t] = pmc(p)
        shutil.copy(p, os.path.join(out, f"{tag}_{t}_counter_collection.csv"))
json.dump(summary, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1)
for a, b in (("trace/kt_kernel_stats.csv", "bench_C2_kernel_stats.csv"), ("bench.json", "bench_C2.json"),
             ("pytest_gpu.log", "pytest_gpu.log"), ("bench_torchrun1.json", "bench_C2_torchrun_1rank.json")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(out, f"{tag}_{b}"))
mk = [k for k in summary.get("pmc_fetch", {}) if k.startswith("void sketch_mask_kernel")]
if mk and "pmc_write" in summary:
    f = sum(summary["pmc_fetch"][k]["FETCH_SIZE"]["avg"] * summary["pmc_fetch"][k]["FETCH_SIZE"]["launches"] for k in mk)
    n = sum(summary["pmc_fetch"][k]["FETCH_SIZE"]["launches"] for k in mk)
    w = sum(summary["pmc_write"][k]["WRITE_SIZE"]["avg"] * summary["pmc_write"][k]["WRITE_SIZE"]["launches"] for k in mk)
    f, w = f / n * 1024, w / n * 1024
    extra = {}
    if "pmc_sq" in summary:
        vi = sum(summary["pmc_sq"][k]["SQ_INSTS_VALU"]["avg"] * summary["pmc_sq"][k]["SQ_INSTS_VALU"]["launches"] for k in mk if k in summary["pmc_sq"])
        vn = sum(summary["pmc_sq"][k]["SQ_INSTS_VALU"]["launches"] for k in mk if k in summary["pmc_sq"])
        if vn:
            extra = {"valu_wave_instr_per_launch": int(vi / vn),
                     "valu_lane_instr_per_base": round(vi / vn * 64 / 275.08e6, 1),  # C2: (500.09 + 50.06) Mbases per step / 2 launches
                     "valu_source": f"profiles/{tag}_pmc_sq_counter_collection.csv SQ_INSTS_VALU, average over the sketch_mask_kernel launches"}
    json.dump({"C2": {**extra, "bytes_per_launch": int(2 * f + w), "fetch_size_bytes": int(f), "write_size_bytes": int(w),
                      "source": f"profiles/{tag}_pmc_fetch/_pmc_write_counter_collection.csv: average over the sketch_mask_kernel "
                                "launches of one bench step (contig + read sketch), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                                "passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B on wide "
                                "coalesced streams; the loads here are dword-wide, so this is an upper bound)"}},
              open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print("traffic per launch:", int(2 * f + w), "fetch", int(f), "write", int(w))

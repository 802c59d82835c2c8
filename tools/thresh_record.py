#!/usr/bin/env python3
"""profiles/<tag>_threshold_pass.json: the window passes side by side on the C3 read launches -- the
threshold pass (default), the block-minima pass (NTL_SKETCH_THRESH=0) and the threshold pass without staged keys
(NTL_SKETCH_THRESH_DIRECT=1) -- from the kernel traces and SQ counter passes of tools/gpu_round3.sh (NTL_PIPELINE=0).
usage: tools/thresh_record.py <tag>"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
KERNELS = ("sketch_thresh_kernel", "sketch_fast_kernel", "sketch_lanes_kernel")


def counters(path):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"], "c": {}, "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    by = collections.defaultdict(list)
    for d in disp.values():
        if any(k in d["k"] for k in KERNELS):
            by[d["k"]].append(d)
    name = max(by, key=lambda k: sum(x["ns"] for x in by[k]))
    return name, by[name][1:]  # the first launch is the contig stage


def trace_avg(path, kernel):
    for r in csv.DictReader(open(path)):
        if r["Name"].split("(")[0] == kernel.split("(")[0]:
            return float(r["AverageNs"]) / 1e6, int(r["Calls"])
    return None, 0


def bases(path):
    for ln in open(path):
        if ln.startswith('{"metric'):
            return json.loads(ln)["roofline"]["bases_per_launch"]


out = {}
for label, sfx, how in (("threshold pass (default)", "", "default"), ("block-minima pass", "_fast", "NTL_SKETCH_THRESH=0"),
                        ("threshold pass without staged keys", "_direct", "NTL_SKETCH_THRESH_DIRECT=1")):
    p = os.path.join(src, f"pmc_sq_C3{sfx}", "p_counter_collection.csv")
    if not os.path.exists(p):
        continue
    name, ks = counters(p)
    n = len(ks)
    avg = lambda c: sum(d["c"][c] for d in ks) / n  # noqa: E731
    cyc = avg("GRBM_GUI_ACTIVE") / 8.0
    bpl = bases(os.path.join(src, f"bench_trace_C3{sfx}.json"))
    t_ms, calls = trace_avg(os.path.join(src, f"trace_C3{sfx}", "kt_kernel_stats.csv"), name)
    out[label] = {"selected_by": how, "kernel": name.split("(")[0], "launches": n, "bases_per_launch": bpl,
                  "unprofiled_trace_avg_ms": round(t_ms, 4) if t_ms else None, "trace_calls": calls,
                  "profiled_launch_ms": round(sum(d["ns"] for d in ks) / n / 1e6, 4),
                  "valu_wave_instr_per_launch": int(avg("SQ_INSTS_VALU")), "valu_lane_instr_per_base": round(avg("SQ_INSTS_VALU") * 64 / bpl, 2),
                  "valu_busy_frac": round(avg("SQ_ACTIVE_INST_VALU") * 4.0 / (1024 * cyc), 4), "lds_instr": int(avg("SQ_INSTS_LDS")),
                  "salu_instr": int(avg("SQ_INSTS_SALU")), "wait_inst_any_over_wave_cycles": round(avg("SQ_WAIT_INST_ANY") / avg("SQ_WAVE_CYCLES"), 3),
                  "clock_ghz": round(cyc / (sum(d["ns"] for d in ks) / n), 3)}
out["what"] = ("VERDICT r2 item 2b, the structural experiment: threshold-sparsified windows (sketch_thresh_kernel, ntlink_amd/csrc/sketch2_kernels.h). "
               f"C3 read launches, NTL_PIPELINE=0, rocprofv3 --kernel-trace --stats and one --pmc pass per variant (gpurun_out/{tag}). "
               "Kill criterion: kept only if the launch gets faster. It does: it is the default window pass for 71 <= w <= 255.")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_threshold_pass.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

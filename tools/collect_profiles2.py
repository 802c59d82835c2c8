#!/usr/bin/env python3
"""Copies the summaries of one GPU visit (gpurun_out/<tag>/, written by tools/gpu_round2.sh) into profiles/ and
refreshes profiles/traffic.json[workload]: HBM bytes and VALU instructions per READ-batch launch of the dominant
sketch kernel, from the PMC passes of the same bench command, together with the kernel-source signature and the
bases per launch the passes were taken on (bench.py quotes them only while both still match).
usage: tools/collect_profiles2.py <tag> [workload]"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_signature only)

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "C3"
src = os.path.join(ROOT, "gpurun_out", tag)
out = os.path.join(ROOT, "profiles")
SKETCH = ("sketch_mask_kernel", "sketch_fast_kernel")


def rows(path):
    """per dispatch: kernel, counters, duration; CSV has one row per (dispatch, counter)"""
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"], "c": {}, "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                                   "grid": int(r["Grid_Size"]), "vgpr": int(r["VGPR_Count"]), "lds": int(r["LDS_Block_Size"])})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(disp.values())


def summarise(rs):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in rs:
        for c, v in d["c"].items():
            agg[d["kernel"].split("(")[0][:72]][c].append(v)
    return {k: {c: {"launches": len(x), "avg": sum(x) / len(x), "max": max(x)} for c, x in v.items()} for k, v in agg.items()}


def read_launches(rs):
    """dispatches of the dominant sketch kernel that belong to read batches: all but the first (the contig stage)"""
    ks = collections.defaultdict(list)
    for d in rs:
        if any(s in d["kernel"] for s in SKETCH) and "true" not in d["kernel"].split("<")[-1].split(",")[2:3]:
            ks[d["kernel"]].append(d)
    if not ks:
        return None, []
    name = max(ks, key=lambda k: sum(x["ns"] for x in ks[k]))
    return name, ks[name][1:]


summary, per = {}, {}
for t in ("pmc_fetch", "pmc_write", "pmc_sq"):
    p = os.path.join(src, t, "p_counter_collection.csv")
    if os.path.exists(p):
        rs = rows(p)
        summary[t] = summarise(rs)
        per[t] = read_launches(rs)
        shutil.copy(p, os.path.join(out, f"{tag}_{t}_counter_collection.csv"))
json.dump(summary, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1)
for a, b in (("trace/kt_kernel_stats.csv", f"bench_{workload}_kernel_stats.csv"), ("bench.json", f"bench_{workload}.json"),
             ("pytest_gpu.log", "pytest_gpu.log"), ("bench_torchrun1.json", f"bench_{workload}_torchrun_1rank.json"),
             ("bench_trace.json", f"bench_{workload}_profiled_run.json")):
    if os.path.exists(os.path.join(src, a)) and os.path.getsize(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(out, f"{tag}_{b}"))

bj = None
for cand in ("bench_trace.json", "bench_pmc_sq.json", "bench.json"):
    p = os.path.join(src, cand)
    if os.path.exists(p):
        for ln in open(p):
            if ln.startswith("{") and '"metric"' in ln:
                bj = json.loads(ln)
        if bj:
            break
if bj and all(t in per and per[t][1] for t in ("pmc_fetch", "pmc_write")):
    avg = lambda t, c: sum(d["c"][c] for d in per[t][1]) / len(per[t][1])
    f, w = avg("pmc_fetch", "FETCH_SIZE") * 1024, avg("pmc_write", "WRITE_SIZE") * 1024
    kname = per["pmc_fetch"][0].split("(")[0]
    ent = {"kernel": kname, "kernel_signature": bench.kernel_signature(), "bases_per_launch": bj["roofline"]["bases_per_launch"],
           "bytes_per_launch": int(2 * f + w), "fetch_size_bytes": int(f), "write_size_bytes": int(w),
           "source": f"profiles/{tag}_pmc_fetch/_pmc_write_counter_collection.csv: average over the {len(per['pmc_fetch'][1])} read-batch launches of "
                     f"{kname} in `bench.py --steps 1 --warmup 1`, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE "
                     "doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B on wide coalesced reads; an upper bound for the dword loads here)"}
    if "pmc_sq" in per and per["pmc_sq"][1]:
        n = len(per["pmc_sq"][1])
        vi = avg("pmc_sq", "SQ_INSTS_VALU")
        ent.update(valu_wave_instr_per_launch=int(vi), valu_lane_instr_per_base=round(vi * 64 / ent["bases_per_launch"], 2),
                   salu_per_valu=round(avg("pmc_sq", "SQ_INSTS_SALU") / vi, 3), lds_per_valu=round(avg("pmc_sq", "SQ_INSTS_LDS") / vi, 3),
                   valu_source=f"profiles/{tag}_pmc_sq_counter_collection.csv SQ_INSTS_VALU, average over the {n} read-batch launches")
        dur = sum(d["ns"] for d in per["pmc_sq"][1]) / n
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS): clock = count / 8 / duration
        ent["clock_ghz"] = round(avg("pmc_sq", "GRBM_GUI_ACTIVE") / 8.0 / dur, 3)
        ent["wait_inst_any_over_wave_cycles"] = round(avg("pmc_sq", "SQ_WAIT_INST_ANY") / max(avg("pmc_sq", "SQ_WAVE_CYCLES"), 1), 3)
    tj = os.path.join(out, "traffic.json")
    cur = json.load(open(tj)) if os.path.exists(tj) else {}
    cur[workload] = ent
    json.dump(cur, open(tj, "w"), indent=1)
    print(json.dumps(ent, indent=1))
else:
    print("no PMC passes or no bench line: traffic.json unchanged")

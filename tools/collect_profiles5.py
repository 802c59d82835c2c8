#!/usr/bin/env python3
"""Copies the summaries of one GPU visit (gpurun_out/<tag>/, written by tools/gpu_round5.sh) into profiles/ and refreshes
profiles/traffic.json[workload] for every workload that was profiled: HBM bytes, VALU instructions and the SIMD cycles
per VALU instruction of a READ-batch launch of the dominant window kernel, and the same counters for the other big kernels of a step
(emit with the index lookup, the map kernels), from the PMC passes (NTL_PIPELINE=0: kernels alone) of the bench command,
together with the kernel-source signature and the bases per launch the passes were taken on (bench.py quotes them only
while both still match).   usage: tools/collect_profiles5.py <tag>"""
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
src = os.path.join(ROOT, "gpurun_out", tag)
out = os.path.join(ROOT, "profiles")
SKETCH = ("sketch_mask_kernel", "sketch_fast_kernel", "sketch_thresh_kernel", "sketch_wave_kernel")
N_SIMD = 1024


def rows(path):
    """per dispatch: kernel, counters, duration; CSV has one row per (dispatch, counter)"""
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"], "c": {}, "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                                   "grid": int(r["Grid_Size"]), "vgpr": int(r["VGPR_Count"]), "lds": int(r["LDS_Block_Size"])})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(disp.values())


def short(k):
    return k.split("(")[0].replace("void ", "")[:72]


def summarise(rs):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    meta = {}
    for d in rs:
        dur[short(d["kernel"])].append(d["ns"])
        meta[short(d["kernel"])] = {"vgpr": d["vgpr"], "lds": d["lds"]}
        for c, v in d["c"].items():
            agg[short(d["kernel"])][c].append(v)
    return {k: dict({c: {"launches": len(x), "avg": sum(x) / len(x), "max": max(x)} for c, x in v.items()},
                    avg_ns=sum(dur[k]) / len(dur[k]), total_ms=sum(dur[k]) / 1e6, **meta[k]) for k, v in agg.items()}


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


for a, b in (("bench.json", "bench_default.json"), ("pytest_gpu.log", "pytest_gpu.log"), ("bench_torchrun1.json", "bench_torchrun_1rank.json"),
             ("smoke.log", "smoke.log")):
    if os.path.exists(os.path.join(src, a)) and os.path.getsize(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(out, f"{tag}_{b}"))

tj = os.path.join(out, "traffic.json")
cur = json.load(open(tj)) if os.path.exists(tj) else {}
for workload in ("C3", "C5", "C2"):
    summary, per, allrows = {}, {}, {}
    for t in ("pmc_fetch", "pmc_write", "pmc_sq"):
        p = os.path.join(src, f"{t}_{workload}", "p_counter_collection.csv")
        if os.path.exists(p):
            rs = rows(p)
            allrows[t] = rs
            summary[t] = summarise(rs)
            per[t] = read_launches(rs)
    if not summary:
        continue
    json.dump(summary, open(os.path.join(out, f"{tag}_pmc_summary_{workload}.json"), "w"), indent=1)
    for a, b in ((f"trace_{workload}/kt_kernel_stats.csv", f"bench_{workload}_serial_kernel_stats.csv"),
                 (f"trace_{workload}_pipe/kt_kernel_stats.csv", f"bench_{workload}_pipelined_kernel_stats.csv"),
                 (f"bench_trace_{workload}_pipe.json", f"bench_{workload}_pipelined_profiled_run.json"),
                 (f"bench_trace_{workload}.json", f"bench_{workload}_profiled_run.json")):
        if os.path.exists(os.path.join(src, a)) and os.path.getsize(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(out, f"{tag}_{b}"))
    bj = None
    for cand in (f"bench_trace_{workload}.json", f"bench_pmc_sq_{workload}.json"):
        p = os.path.join(src, cand)
        if os.path.exists(p):
            for ln in open(p):
                if ln.startswith("{") and '"metric"' in ln:
                    bj = json.loads(ln)
            if bj:
                break
    if not (bj and all(t in per and per[t][1] for t in ("pmc_fetch", "pmc_write"))):
        print(workload, ": no PMC passes or no bench line: traffic.json entry unchanged")
        continue
    avg = lambda t, c: sum(d["c"][c] for d in per[t][1]) / len(per[t][1])  # noqa: E731
    f, w = avg("pmc_fetch", "FETCH_SIZE") * 1024, avg("pmc_write", "WRITE_SIZE") * 1024
    kname = per["pmc_fetch"][0].split("(")[0]
    ent = {"kernel": kname, "kernel_signature": bench.kernel_signature(), "bases_per_launch": bj["roofline"]["bases_per_launch"],
           "bytes_per_launch": int(2 * f + w), "fetch_size_bytes": int(f), "write_size_bytes": int(w),
           "source": f"profiles/{tag}_pmc_summary_{workload}.json (gpurun_out/{tag}/pmc_fetch_{workload}, pmc_write_{workload}): average over the {len(per['pmc_fetch'][1])} read-batch launches of "
                     f"{kname} in `NTL_PIPELINE=0 bench.py --workload {workload} --steps 1 --warmup 1`, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE "
                     "doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B on wide coalesced reads; an upper bound for the dword loads here)",
           }
    if "pmc_sq" in per and per["pmc_sq"][1]:
        n = len(per["pmc_sq"][1])
        vi = avg("pmc_sq", "SQ_INSTS_VALU")
        dur = sum(d["ns"] for d in per["pmc_sq"][1]) / n
        cyc = avg("pmc_sq", "GRBM_GUI_ACTIVE") / 8.0  # summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS)
        simd_cycles = N_SIMD * cyc
        ent.update(valu_wave_instr_per_launch=int(vi), valu_lane_instr_per_base=round(vi * 64 / ent["bases_per_launch"], 2),
                   simd_cycles_per_launch=int(simd_cycles), measured_cycles_per_wave_instr=round(simd_cycles / vi, 3),
                   salu_per_valu=round(avg("pmc_sq", "SQ_INSTS_SALU") / vi, 3), lds_per_valu=round(avg("pmc_sq", "SQ_INSTS_LDS") / vi, 3),
                   clock_ghz=round(cyc / dur, 3), profiled_launch_ms=round(dur / 1e6, 4),
                   **({"strips_per_launch": round(bj["config"]["window_strips_per_step"] / max(bj.get("sub_batches_per_rank") or 1, 1), 1),
                       "per_strip": {c[9:].lower(): round(avg("pmc_sq", c) / (bj["config"]["window_strips_per_step"] / max(bj.get("sub_batches_per_rank") or 1, 1)), 1)
                                     for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU")}}
                      if bj.get("config", {}).get("window_strips_per_step") else {}),
                   wait_inst_any_over_wave_cycles=round(avg("pmc_sq", "SQ_WAIT_INST_ANY") / max(avg("pmc_sq", "SQ_WAVE_CYCLES"), 1), 3),
                   isa_mix=os.environ.get("NTL_ISA_MIX", "profiles/r06_isa_mix.json"),
                   valu_source=f"profiles/{tag}_pmc_summary_{workload}.json (pmc_sq pass): SQ_INSTS_VALU and GRBM_GUI_ACTIVE (/ 8 XCDs x 1024 SIMDs = SIMD cycles), "
                               f"average over the {n} read-batch launches (SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU in these files: it is not a busy-cycle count and is not used)")
    # the other big kernels of a step: bytes, VALU share, duration per launch
    others = {}
    for k in set(summary.get("pmc_sq", {})) | set(summary.get("pmc_fetch", {})):
        if not any(x in k for x in ("emit_kernel", "emit_list_kernel", "map_kernel", "map_overflow", "map_gather", "mask_count", "probe_kernel", "sketch_fast_list")):
            continue
        e = {}
        sq = summary.get("pmc_sq", {}).get(k)
        if sq and "SQ_INSTS_VALU" in sq:
            cyc_k = sq["GRBM_GUI_ACTIVE"]["avg"] / 8.0
            e.update(launches=sq["SQ_INSTS_VALU"]["launches"], avg_ms=round(sq["avg_ns"] / 1e6, 4), vgpr=sq["vgpr"], lds=sq["lds"],
                     valu_wave_instr=int(sq["SQ_INSTS_VALU"]["avg"]),
                     simd_cycles_per_valu_instr=round(N_SIMD * cyc_k / max(sq["SQ_INSTS_VALU"]["avg"], 1), 2))
        fe, wr = summary.get("pmc_fetch", {}).get(k), summary.get("pmc_write", {}).get(k)
        if fe and wr and "FETCH_SIZE" in fe and "WRITE_SIZE" in wr:
            e.update(fetch_bytes=int(fe["FETCH_SIZE"]["avg"] * 1024), write_bytes=int(wr["WRITE_SIZE"]["avg"] * 1024),
                     hbm_bytes_2f_plus_w=int(2 * fe["FETCH_SIZE"]["avg"] * 1024 + wr["WRITE_SIZE"]["avg"] * 1024))
        others[k] = e
    ent["other_kernels"] = others
    cur[workload] = ent
    print(workload, json.dumps({k: v for k, v in ent.items() if k != "other_kernels"}, indent=1))
    for k, v in others.items():
        print("   ", k, v)
json.dump(cur, open(tj, "w"), indent=1)

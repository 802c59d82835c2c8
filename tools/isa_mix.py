#!/usr/bin/env python3
"""VALU instruction mix of one kernel BY LOOP DEPTH, priced in issue cycles (the class-priced roof from the
dynamic mix, not the static one).

The compiler's assembly says for every basic block which loop it belongs to and how deep that loop is nested
(`; in Loop: Header=BBx_y Depth=N`, `; =>This Inner Loop Header: Depth=N`).  For sketch_wave_kernel depth 1 is the loop over the
strips a wavefront takes (the 64 rolling steps are straight-line code inside it), depth 2 the scan rounds of a strip, depth 3 the
four-entries-per-step scans.  Every VALU mnemonic is put into an issue class of profiles/valu_cycles.json (SIMD cycles per wave64
instruction, measured with tools/valu_calib*.hip), and the output holds, per depth, instruction counts and priced cycles:
bench.py weights the depths by trip counts -- depth 1 once per strip, depth 2 by the rounds per strip, depth 3 fitted so that the
total equals the SQ_INSTS_VALU the profiler counted -- and gets the cycles per wave-instruction this kernel WOULD take if every
instruction issued at its calibrated cost.

usage: tools/isa_mix.py <kernel-name-regex> [--asm file.s] [-o profiles/r04_isa_mix.json]"""
import argparse
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hist  # noqa: E402


def price_table():
    d = json.load(open(isa_hist.CALIB))
    cyc, fam = d["cycles"], [(re.compile(p), c) for p, c in d["families"]]

    def price(mn):
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn)
        if base in cyc:
            return base, cyc[base]
        for rx, cls in fam:
            if rx.search(base):
                return cls, cyc[cls]
        return "other_valu", cyc["other_valu"]
    return price


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--asm")
    ap.add_argument("-o", "--out")
    a = ap.parse_args()
    asm = isa_hist.device_asm(a.asm)
    hsa = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, re.M))
    names = isa_hist.demangle(sorted(hsa))
    rx = re.compile(a.kernel)
    price = price_table()
    out = {"source": "hipcc --offload-arch=gfx950 -O3 --offload-device-only -S ntl_hip.hip; loop depths from the compiler's block comments",
           "cycles_table": "profiles/valu_cycles.json", "kernels": {}}
    cur, depth = None, 0
    per = None
    for ln in asm.splitlines():
        m = re.match(r"^(\S+):", ln)
        if m and not ln.startswith("."):
            cur = m.group(1) if m.group(1) in hsa and rx.search(names[m.group(1)]) else None
            if cur:
                per = out["kernels"].setdefault(names[cur], {"by_depth": collections.defaultdict(lambda: {"valu": 0, "priced_cycles": 0.0, "salu": 0, "lds": 0, "vmem": 0,
                                                                                              "classes": collections.Counter()})})["by_depth"]
                depth = 0
            continue
        if cur is None:
            continue
        t = ln.strip()
        if t.startswith(".Lfunc_end"):
            cur = None
            continue
        if re.match(r"^\.?LBB\d+_\d+:", t) or t.startswith("; %bb."):
            d = re.search(r"in Loop: Header=\S+ Depth=(\d+)", t)
            depth = int(d.group(1)) if d else 0  # a loop header says its own depth on a comment line below its label
            continue
        d = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", t)
        if d:
            depth = int(d.group(1))
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        mn = t.split()[0]
        e = per[depth]
        if mn.startswith("v_"):
            cls, c = price(mn)
            e["valu"] += 1
            e["priced_cycles"] += c
            e["classes"][cls] += 1
        elif mn.startswith("s_"):
            e["salu"] += 1
        elif mn.startswith("ds_"):
            e["lds"] += 1
        elif mn.startswith(("global_", "buffer_", "flat_")):
            e["vmem"] += 1
    for k in out["kernels"].values():
        k["by_depth"] = {str(d): {**v, "priced_cycles": round(v["priced_cycles"], 1), "classes": dict(v["classes"].most_common())} for d, v in sorted(k["by_depth"].items())}
    txt = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    for nm, k in out["kernels"].items():
        print(nm)
        for d, v in k["by_depth"].items():
            print(f"  depth {d}: {v['valu']} VALU = {v['priced_cycles']} cycles ({v['priced_cycles'] / max(v['valu'], 1):.2f}/instr), {v['salu']} SALU, {v['lds']} LDS, {v['vmem']} VMEM")


if __name__ == "__main__":
    main()

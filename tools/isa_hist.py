#!/usr/bin/env python3
"""Static instruction histogram of the shipped gfx950 kernels.

Compiles ntlink_amd/csrc/ntl_hip.hip to device assembly with the flags of ntlink_amd/build.py (`hipcc --offload-arch=gfx950
-O3 --offload-device-only -S`; no GPU needed) and counts mnemonics per kernel.  Every VALU mnemonic is put into an issue class
whose cost (SIMD cycles per wave64 instruction at 8 waves per SIMD) comes from tools/valu_calib2.hip's measurements
(profiles/r02a_valu_calib2.txt + the round-3 additions); the static mix gives `cycles_per_valu_instr` = sum(count x cycles) /
sum(count), the denominator of bench.py's VALU roof.  usage: tools/isa_hist.py [kernel-name-regex] [-o out.json] [--asm file.s]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALIB = os.path.join(ROOT, "profiles", "valu_cycles.json")


def device_asm(path=None):
    if path:
        return open(path).read()
    out = os.path.join(tempfile.mkdtemp(prefix="ntl_isa_"), "ntl_hip.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "-I", os.path.join(ROOT, "ntlink_amd", "csrc"),
           "--offload-device-only", "-S", os.path.join(ROOT, "ntlink_amd", "csrc", "ntl_hip.hip"), "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return open(out).read()


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return dict(zip(names, p.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def kernels(asm):
    """{mangled name: [mnemonic, ...]} for every .amdhsa_kernel of the file"""
    lines = asm.splitlines()
    hsa = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, re.M))
    out, cur = {}, None
    for ln in lines:
        m = re.match(r"^(\S+):\s", ln + " ")
        if m and not ln.startswith("."):
            cur = m.group(1) if m.group(1) in hsa else None
            if cur:
                out[cur] = []
            continue
        if cur is None:
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            if t.startswith(".Lfunc_end"):
                cur = None
            continue
        mn = t.split()[0]
        if re.match(r"^[a-z_0-9]+$", mn):
            out[cur].append(mn)
    return out


def classify(mn, table):
    """issue class of a VALU mnemonic: exact entry of the calibration table, else its family default"""
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", mn)
    if base in table["cycles"]:
        return base, table["cycles"][base]
    for pat, cls in table["families"]:
        if re.match(pat, base):
            return cls, table["cycles"][cls]
    return "other_valu", table["cycles"]["other_valu"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pattern", nargs="?", default=r"sketch_thresh_kernel<256, (false|true)>|sketch_fast_list_kernel<256, 10>|sketch_fast_kernel<256, 10, false>|sketch_fast_kernel<256, 4, false>|emit_kernel<[12], 1>|map_kernel<512, 128, 1>|map_kernel<256, 64, 0>|map_kernel<1024, 128, 2>")
    ap.add_argument("-o", "--out", default=None)
    ap.add_argument("--asm", default=None)
    a = ap.parse_args()
    table = json.load(open(CALIB))
    ks = kernels(device_asm(a.asm))
    names = demangle(list(ks))
    res = {}
    for mangled, mns in ks.items():
        nm = names[mangled]
        if not re.search(a.pattern, nm):
            continue
        hist = collections.Counter(mns)
        valu = {m: c for m, c in hist.items() if m.startswith("v_") and not m.startswith("v_mfma")}
        classes = collections.Counter()
        cyc = 0.0
        for m, c in valu.items():
            cls, cy = classify(m, table)
            classes[f"{cls} ({cy})"] += c
            cyc += c * cy
        nv = sum(valu.values())
        res[nm.replace("void ", "").split("(")[0]] = {
            "instructions": len(mns), "valu": nv, "salu": sum(c for m, c in hist.items() if m.startswith("s_")),
            "lds": sum(c for m, c in hist.items() if m.startswith("ds_")),
            "vmem": sum(c for m, c in hist.items() if m.startswith(("global_", "flat_", "buffer_", "scratch_"))),
            "cycles_per_valu_instr_static_mix": round(cyc / nv, 3) if nv else None,
            "valu_by_class": dict(classes.most_common()),
            "valu_by_mnemonic": dict(sorted(valu.items(), key=lambda kv: -kv[1])),
        }
    txt = json.dumps({"source": "hipcc --offload-arch=gfx950 -O3 --offload-device-only -S ntl_hip.hip (static counts; loops are counted once)",
                      "cycles_table": os.path.relpath(CALIB, ROOT), "kernels": res}, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    else:
        print(txt)


if __name__ == "__main__":
    main()

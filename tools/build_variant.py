#!/usr/bin/env python3
"""Builds a VARIANT of the kernel unit here (no GPU needed): ntl_hip.hip with extra hipcc flags -> ntlink_amd/build/var_<name>/libntlink_hip.so,
linked with the default build's host objects.  The variants travel to the GPU box with the snapshot (build/ is git-ignored, not
gpurun-ignored) and are selected with NTLINK_AMD_LIB (capi.load): an A/B costs no compile time on the box.
usage: tools/build_variant.py <name> "<extra flags>" [<name> "<flags>" ...]   (built in parallel)"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ntlink_amd import build as B  # noqa: E402


def one(name, flags):
    d = os.path.join(B.OBJ, "var_" + name)
    os.makedirs(d, exist_ok=True)
    obj = os.path.join(d, "ntl_hip.o")
    hipcc = B.hipcc_path()
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-I", B.CSRC, *flags.split(),
                           "-c", os.path.join(B.CSRC, "ntl_hip.hip"), "-o", obj])
    others = [os.path.join(B.OBJ, u.rsplit(".", 1)[0] + ".o") for u in B.UNITS if u != "ntl_hip.hip"]
    out = os.path.join(d, "libntlink_hip.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-lz", "-ldl", "-lpthread", "-o", out])
    open(os.path.join(d, "flags.txt"), "w").write(flags)
    return out


if __name__ == "__main__":
    B.build_hip()
    pairs = list(zip(sys.argv[1::2], sys.argv[2::2]))
    with ThreadPoolExecutor(max(1, min(4, len(pairs)))) as ex:
        for p in ex.map(lambda nf: one(*nf), pairs):
            print(p)

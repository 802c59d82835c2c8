"""Builds libntlink_hip.so (the HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libntlink_hip.so")
SOURCES = ["ntl_hip.hip", "ntl_io.cpp", "ntl_pairs.cpp", "dev_common.h", "dev_intrin.h", "scan_kernels.h", "sketch_kernels.h", "sketch2_kernels.h", "map_kernels.h", "pack_kernels.h", "synth_kernels.h"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_hip(force=False, extra_flags=()):
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(HERE), "include", "ntlink_amd.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs):
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result",
           "-I", CSRC, *extra_flags, os.path.join(CSRC, "ntl_hip.hip"), os.path.join(CSRC, "ntl_io.cpp"), os.path.join(CSRC, "ntl_pairs.cpp"), "-lz", "-ldl", "-lpthread", "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_hip(force=True))

"""Builds and opens the SIMT-mock build of the product's kernels (test tool, see include/hip/hip_runtime.h)."""
import os
import subprocess

from ntlink_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SIM_LIB = os.path.join(HERE, "build", "libntlink_sim.so")
CSRC = os.path.join(ROOT, "ntlink_amd", "csrc")
UNITS = [os.path.join(CSRC, f) for f in ("ntl_hip.hip", "ntl_io.cpp", "ntl_pairs.cpp", "ntl_liftover.cpp")]
# every header of the product's kernels is a dependency (dev_intrin.h is replaced by the mock's own file of that name)
SRC = UNITS + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
      [os.path.join(HERE, "sim_runtime.cpp"), os.path.join(HERE, "include", "hip", "hip_runtime.h"),
       os.path.join(HERE, "include", "dev_intrin.h")]


def build(sanitize=False):
    """Serialised across processes (pytest-xdist workers, the ranks of the multi-process tests): a file lock around the build."""
    import fcntl
    os.makedirs(os.path.dirname(SIM_LIB), exist_ok=True)
    with open(os.path.join(os.path.dirname(SIM_LIB), ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_locked(sanitize)


def _build_locked(sanitize):
    out = SIM_LIB if not sanitize else SIM_LIB.replace(".so", "_asan.so")
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in SRC):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["g++", "-O1", "-g1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-Wall",
           "-Wno-unused-function", "-Wno-unknown-pragmas", "-x", "c++",
           "-I", os.path.join(HERE, "include"), "-I", os.path.join(ROOT, "ntlink_amd", "csrc"),
           *UNITS, os.path.join(HERE, "sim_runtime.cpp"), "-lz", "-ldl", "-o", out]
    if sanitize:
        cmd[1:1] = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    subprocess.check_call(cmd)
    return out


def device():
    return capi.Device(0, lib_path=build())

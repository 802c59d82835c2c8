"""Builds libntlink_hip.so (the HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

One object per translation unit under ntlink_amd/build/ (rebuilt when the unit or a header it includes is newer), then
one link: a change in the host-side I/O code does not recompile the kernels."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libntlink_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "ntlink_amd.h")
# every header under csrc/ is a dependency of the kernel unit (a list kept by hand once missed three of them: an edit of
# index_common.h did not rebuild ntl_hip.o, and objects travel to the GPU box)
KERNEL_HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))
# translation unit -> headers it depends on (besides include/ntlink_amd.h)
UNITS = {"ntl_hip.hip": KERNEL_HEADERS, "ntl_io.cpp": [], "ntl_pairs.cpp": [], "ntl_liftover.cpp": []}
SOURCES = list(UNITS) + KERNEL_HEADERS


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)


def build_hip(force=False, extra_flags=()):
    """Serialised across processes (several test workers may ask for the library at once): a file lock around the build."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_hip_locked(force, extra_flags)


def _build_hip_locked(force, extra_flags):
    hipcc = hipcc_path()
    # NTL_EXTRA_HIPCC_FLAGS: tools only (e.g. -DNTL_SKETCH_ABLATION for tools/gpu_ablate.sh)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-I", CSRC, *extra_flags,
             *os.environ.get("NTL_EXTRA_HIPCC_FLAGS", "").split()]
    tag = os.path.join(OBJ, "flags.txt")
    if not os.path.exists(tag) or open(tag).read() != " ".join(flags):
        force = True
    jobs, objs = [], []
    for unit, hdrs in UNITS.items():
        src = os.path.join(CSRC, unit)
        obj = os.path.join(OBJ, unit.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        deps = [src, HEADER] + [os.path.join(CSRC, h) for h in hdrs if os.path.exists(os.path.join(CSRC, h))]
        if force or _stale(obj, deps):
            jobs.append([hipcc, *flags, "-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(len(jobs)) as ex:
            list(ex.map(subprocess.check_call, jobs))
        open(tag, "w").write(" ".join(flags))
    if jobs or _stale(OUT, objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lz", "-ldl", "-lpthread", "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build_hip(force=True))

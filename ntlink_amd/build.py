"""Builds libntlink_hip.so (the HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

One object per translation unit under ntlink_amd/build/, then one link: a change in the host-side I/O code does not recompile the
kernels.  Staleness is decided by CONTENT (round 6), not by modification times: every object carries the sha256 of its unit, the headers
it depends on and the flags (`<obj>.sig`), the library the sha256 of all of them (`libntlink_hip.so.sig`); objects and the library travel
to the GPU box with the snapshot, and `capi.load` refuses a default library whose signature is not that of the sources beside it --
what runs is what is committed.  `build_hip` says on stderr whether it rebuilt or reused."""
import hashlib
import os
import shutil
import subprocess
import sys
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


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _flags(extra_flags=()):
    # NTL_EXTRA_HIPCC_FLAGS: tools only (e.g. -DNTL_SKETCH_ABLATION for tools/gpu_ablate.sh).  The include path is written relative to
    # the signature: the same sources under another root are the same build
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", *extra_flags,
            *os.environ.get("NTL_EXTRA_HIPCC_FLAGS", "").split()]


def _unit_deps(unit):
    return [os.path.join(CSRC, unit), HEADER] + [os.path.join(CSRC, h) for h in UNITS[unit] if os.path.exists(os.path.join(CSRC, h))]


def source_signature(extra_flags=()):
    """sha256 over every source of the library and the compiler flags: what `libntlink_hip.so.sig` must hold for the library to be current."""
    flags = " ".join(_flags(extra_flags))
    return _sha(sorted({d for u in UNITS for d in _unit_deps(u)}), flags)


def _read(path):
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def library_is_current(extra_flags=()):
    return os.path.exists(OUT) and _read(OUT + ".sig") == source_signature(extra_flags)


last_action = None  # "rebuilt" | "reused": what the last build_hip() of this process did


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _flags(extra_flags=()):
    # NTL_EXTRA_HIPCC_FLAGS: tools only (e.g. -DNTL_SKETCH_ABLATION for tools/gpu_ablate.sh).  The include path is written relative to
    # the signature: the same sources under another root are the same build
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", *extra_flags,
            *os.environ.get("NTL_EXTRA_HIPCC_FLAGS", "").split()]


def _unit_deps(unit):
    return [os.path.join(CSRC, unit), HEADER] + [os.path.join(CSRC, h) for h in UNITS[unit] if os.path.exists(os.path.join(CSRC, h))]


def source_signature(extra_flags=()):
    """sha256 over every source of the library and the compiler flags: what `libntlink_hip.so.sig` must hold for the library to be current."""
    flags = " ".join(_flags(extra_flags))
    return _sha(sorted({d for u in UNITS for d in _unit_deps(u)}), flags)


def _read(path):
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def library_is_current(extra_flags=()):
    return os.path.exists(OUT) and _read(OUT + ".sig") == source_signature(extra_flags)


last_action = None  # "rebuilt" | "reused": what the last build_hip() of this process did


def build_hip(force=False, extra_flags=()):
    """Serialised across processes (several test workers may ask for the library at once): a file lock around the build."""
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_hip_locked(force, extra_flags)


def _build_hip_locked(force, extra_flags):
    global last_action
    hipcc = hipcc_path()
    flags = _flags(extra_flags)
    jobs, objs, sigs = [], [], {}
    for unit in UNITS:
        src = os.path.join(CSRC, unit)
        obj = os.path.join(OBJ, unit.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        sigs[obj] = _sha(_unit_deps(unit), " ".join(flags))
        if force or not os.path.exists(obj) or _read(obj + ".sig") != sigs[obj]:
            jobs.append((obj, [hipcc, *flags, "-I", CSRC, "-c", src, "-o", obj]))
    if jobs:
        with ThreadPoolExecutor(len(jobs)) as ex:
            list(ex.map(subprocess.check_call, [j[1] for j in jobs]))
        for obj, _ in jobs:
            with open(obj + ".sig", "w") as fh:
                fh.write(sigs[obj])
    want = source_signature(extra_flags)
    if jobs or not os.path.exists(OUT) or _read(OUT + ".sig") != want:
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lz", "-ldl", "-lpthread", "-o", OUT])
        with open(OUT + ".sig", "w") as fh:
            fh.write(want)
        last_action = "rebuilt"
        what = ", ".join(os.path.basename(j[0]) for j in jobs) or "link only"
    else:
        last_action = "reused"
        what = "objects and library carry the sources' signature"
    print(f"[ntlink_amd.build] libntlink_hip.so {last_action} ({what}; signature {want[:16]})", file=sys.stderr)
    return OUT


if __name__ == "__main__":
    print(build_hip(force=True))

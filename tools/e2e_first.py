#!/usr/bin/env python3
"""Where does a process's FIRST file-to-file pass differ from its second?  The bench's end-to-end input (C3 assembly + --bases of
reads as plain FASTA in /dev/shm), pipeline.run_pair twice with the driver's timeline (NTL_PIPE_TRACE), per-batch device-stage
durations of both runs side by side.  usage: tools/e2e_first.py [--bases 16e9]"""
import argparse
import json
import os
import resource
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NTL_PIPE_TRACE"] = os.path.abspath(os.environ.get("NTL_PIPE_TRACE", "/tmp/ntl_pipe_trace.tsv"))  # (run_pair is called from the data directory)
import bench  # noqa: E402
from ntlink_amd import capi, pipeline, synth  # noqa: E402

VM_KEYS = ("pgfault", "pgmajfault", "pgalloc_normal", "pgfree", "thp_fault_alloc", "thp_fault_fallback", "thp_collapse_alloc", "compact_stall",
           "allocstall_normal", "pgscan_direct", "pgsteal_direct", "pgscan_kswapd", "numa_hit", "numa_miss", "numa_local", "numa_other",
           "pgmigrate_success", "numa_pages_migrated", "numa_hint_faults", "thp_split_page", "pglazyfree", "pgactivate", "pgdeactivate")


def vmstat():
    out = {}
    try:
        for ln in open("/proc/vmstat"):
            k, v = ln.split()
            if k in VM_KEYS:
                out[k] = int(v)
    except OSError:
        pass
    return out


ap = argparse.ArgumentParser()
ap.add_argument("--bases", type=float, default=16e9)
ap.add_argument("--workload", default="C3")
a = ap.parse_args()
dev = capi.Device(0)
wl = synth.DeviceWorkload(dev, a.workload, with_reads=False)
W = wl.W
d = tempfile.mkdtemp(prefix="ntl_e2e_", dir="/dev/shm")
try:
    cbuf, coff = wl.contigs.download()
    bench.write_fasta(os.path.join(d, "asm.fa"), cbuf, coff, b"ctg")
    files = []
    nb = max(1, int(a.bases // 3.95e9))
    for b in range(nb):
        rb, _ = wl.make_reads(int(a.bases) // nb, seed=(77, b))
        rbuf, roff = rb.download()
        rb.close()
        p = os.path.join(d, f"reads_{b:02d}.fa")
        bench.write_fasta(p, rbuf, roff, b"r%d_" % b)
        files.append(os.path.basename(p))
    os.chdir(d)
    out = []
    for run in range(3):
        for f in os.listdir(d):
            if f.startswith("asm.fa."):
                os.remove(os.path.join(d, f))
        pipeline._TRACE.clear()
        vm0 = vmstat()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        st = pipeline.run_pair(dev, "asm.fa", " ".join(files), k=W["k"], w=W["w"], paf=True, pairs_tsv=True, sensitive=W["sensitive"])
        dt = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        vm1 = vmstat()
        rusage = {"user_s": round(ru1.ru_utime - ru0.ru_utime, 3), "sys_s": round(ru1.ru_stime - ru0.ru_stime, 3),
                  "minor_faults": ru1.ru_minflt - ru0.ru_minflt, "voluntary_switches": ru1.ru_nvcsw - ru0.ru_nvcsw,
                  "involuntary_switches": ru1.ru_nivcsw - ru0.ru_nivcsw,
                  "vmstat_host_wide": {k: vm1[k] - vm0[k] for k in vm1 if vm1[k] != vm0.get(k, 0)}}
        shutil.copyfile(os.environ["NTL_PIPE_TRACE"], os.environ["NTL_PIPE_TRACE"] + f".run{run}")
        ev = {}
        for ln in open(os.environ["NTL_PIPE_TRACE"]):
            if ln.startswith("#"):
                continue
            t, th, what, seq = ln.rstrip("\n").split("\t")
            ev.setdefault(what, []).append((float(t), th, int(seq)))
        starts = {s: t for t, _, s in ev.get("dev_start", [])}
        dones = {s: t for t, _, s in ev.get("dev_done", [])}
        per_batch = [round(dones[s] - starts[s], 3) for s in sorted(starts) if s in dones]
        out.append({"run": run, "seconds": round(dt, 3), "rusage": rusage, "first_dev_start": round(min(starts.values()), 3) if starts else None,
                    "last_dev_done": round(max(dones.values()), 3) if dones else None, "device_stage_s_per_batch": per_batch,
                    "t_contigs": st["t_contigs"], "t_contigs_parts": st.get("t_contigs_parts"), "t_device_parts": st.get("t_device_parts"),
                    "t_ingest": round(st["t_ingest"], 3), "t_write": round(st.get("t_write", 0), 3), "reader": st.get("reader"),
                    "first_reader_out": round(ev["reader_out"][0][0], 3) if ev.get("reader_out") else None})
    print(json.dumps(out))
    print("pinned allocations (MB):", [round(c / 1e6, 1) for c in dev.pin_alloc_caps], file=sys.stderr)
    for wk in dev.workers(2):
        print("worker pinned allocations (MB):", [round(c / 1e6, 1) for c in wk.pin_alloc_caps], file=sys.stderr)
finally:
    os.chdir(ROOT)
    shutil.rmtree(d, ignore_errors=True)
    dev.close()

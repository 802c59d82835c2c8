#!/usr/bin/env python3
"""End-to-end diagnostics on the GPU box: the C3 assembly + N Gbases of reads written once as plain FASTA into /dev/shm
(generated on the device), then `pipeline.run_pair` file to file under several settings, each in its own process (the
native reader reads its environment once).  usage: tools/e2e_diag.py [--bases 16e9] [--forms] [--keep DIR]
       tools/e2e_diag.py --run DIR [--batch N]      (one timed run on prepared files; what the parent starts)"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def prepare(d, bases, workload):
    import bench
    from ntlink_amd import capi, synth
    dev = capi.Device(0)
    W = synth.workload(workload, 1.0)
    wl = synth.DeviceWorkload(dev, workload, 1.0, with_reads=False)
    cbuf, coff = wl.contigs.download()
    bench.write_fasta(os.path.join(d, "asm.fa"), cbuf, coff, b"ctg")
    del cbuf
    files, per = [], 3_950_000_000
    nb = max(1, -(-int(bases) // per))
    for b in range(nb):
        rb, _ = wl.make_reads(int(bases) // nb, seed=(77, b))
        rbuf, roff = rb.download()
        rb.close()
        p = os.path.join(d, f"reads_{b:02d}.fa")
        bench.write_fasta(p, rbuf, roff, b"r%d_" % b)
        files.append(os.path.basename(p))
        del rbuf
    wl.close()
    dev.close()
    json.dump({"files": files, "k": W["k"], "w": W["w"], "sensitive": W["sensitive"]}, open(os.path.join(d, "plan.json"), "w"))
    return files


def run_once(d, batch, files=None, repeat=1):
    from ntlink_amd import capi, pipeline
    plan = json.load(open(os.path.join(d, "plan.json")))
    files = files or plan["files"]
    dev = capi.Device(0)
    os.chdir(d)
    keep = ("t_contigs", "t_contigs_parts", "t_ingest", "t_device", "t_device_parts", "t_handover", "t_drain_tail", "t_graph", "t_write", "t_tally", "reader")
    for rep in range(repeat):  # the second and later passes of one process: staging buffers and buffer pools exist (the steady state)
        for f in os.listdir(d):
            if f.startswith("asm.fa."):
                os.remove(f)
        t0 = time.perf_counter()
        st = pipeline.run_pair(dev, "asm.fa", " ".join(files), k=plan["k"], w=plan["w"], paf=True, pairs_tsv=True, sensitive=plan["sensitive"],
                               **({"batch_bases": batch} if batch else {}))  # 0: the driver's default
        dt = time.perf_counter() - t0
        print(json.dumps({"Gbases_per_s": round(st["read_bases"] / dt / 1e9, 3), "seconds": round(dt, 3), "pass": rep, "read_bases": st["read_bases"],
                          "batch_bases": batch, "env": {k: v for k, v in os.environ.items() if k.startswith("NTL_")},
                          **{k: (round(st[k], 4) if isinstance(st.get(k), float) else st.get(k)) for k in keep}}), flush=True)
    dev.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=float, default=16e9)
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--run", default=None)
    ap.add_argument("--batch", type=int, default=256_000_000)
    ap.add_argument("--files", default=None)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--forms", action="store_true", help="also time gzip / FASTQ.gz forms of (a part of) the same reads")
    a = ap.parse_args()
    if a.run:
        run_once(a.run, a.batch, a.files.split(",") if a.files else None, a.repeat)
        return
    import tempfile
    d = tempfile.mkdtemp(prefix="ntl_e2ed_", dir="/dev/shm")
    try:
        t0 = time.perf_counter()
        files = prepare(d, a.bases, a.workload)
        print(json.dumps({"prepared": len(files), "seconds": round(time.perf_counter() - t0, 1)}), flush=True)
        if os.environ.get("NTL_E2E_TRACE_ONLY"):
            for i in range(2):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--run", d, "--batch", "256000000"],
                               env=dict(os.environ, NTL_PIPE_TRACE=os.environ["NTL_E2E_TRACE_ONLY"] + f".{i}", NTL_IO_TRACE="1"), check=False)
            return
        if os.environ.get("NTL_E2E_SWEEP5"):  # the defaults, twice
            for _ in range(2):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--run", d, "--batch", "0", "--repeat", "3"], env=dict(os.environ), check=False)
            return
        if os.environ.get("NTL_E2E_SWEEP4"):
            for env, batch in (({}, 512_000_000), ({"NTL_IO_THREADS": "24"}, 512_000_000), ({"NTL_IO_THREADS": "20"}, 512_000_000), ({}, 1_000_000_000),
                               ({"NTL_IO_THREADS": "24"}, 1_000_000_000), ({"NTL_IO_THREADS": "24"}, 384_000_000), ({"NTL_IO_THREADS": "28"}, 512_000_000),
                               ({"NTL_IO_THREADS": "24", "NTL_IO_READERS": "3"}, 512_000_000)):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--run", d, "--batch", str(batch), "--repeat", "3"], env=dict(os.environ, **env), check=False)
            return
        if os.environ.get("NTL_E2E_SWEEP3"):  # three passes per process: the last ones are the steady state
            for env, batch in (({}, 256_000_000), ({"NTL_IO_READERS": "3"}, 256_000_000), ({"NTL_IO_THREADS": "16"}, 256_000_000),
                               ({"NTL_IO_THREADS": "24"}, 256_000_000), ({"NTL_DEVICE_STREAMS": "3"}, 256_000_000), ({}, 512_000_000),
                               ({"NTL_IO_READERS": "3", "NTL_DEVICE_STREAMS": "3"}, 256_000_000), ({"NTL_IO_READERS": "1"}, 256_000_000)):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--run", d, "--batch", str(batch), "--repeat", "3"], env=dict(os.environ, **env), check=False)
            return
        if os.environ.get("NTL_E2E_SWEEP2"):
            for env in ({}, {}, {"NTL_IO_ONE_PASS": "0"}, {"NTL_IO_THREADS": "8"}, {"NTL_IO_THREADS": "16"}, {"NTL_IO_THREADS": "16", "NTL_IO_READERS": "1"},
                        {"NTL_IO_THREADS": "16", "NTL_IO_READERS": "3"}, {"NTL_IO_THREADS": "24"}, {"NTL_IO_THREADS": "64"}):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--run", d, "--batch", "256000000"], env=dict(os.environ, **env), check=False)
            return
        configs = [({"NTL_IO_READERS": "1"}, 256_000_000), ({"NTL_IO_READERS": "1"}, 256_000_000),  # twice: page-cache and pool warm-up
                   ({"NTL_IO_READERS": "2"}, 256_000_000), ({"NTL_IO_READERS": "3"}, 256_000_000), ({"NTL_IO_READERS": "4"}, 256_000_000),
                   ({"NTL_IO_READERS": "3", "NTL_IO_THREADS": "64"}, 256_000_000),
                   ({"NTL_IO_READERS": "3", "NTL_IO_THREADS": "16"}, 256_000_000),
                   ({"NTL_IO_READERS": "6", "NTL_IO_THREADS": "16"}, 256_000_000),
                   ({"NTL_IO_READERS": "3"}, 512_000_000),
                   ({"NTL_IO_READERS": "3", "NTL_DEVICE_STREAMS": "1"}, 256_000_000),
                   ({"NTL_IO_READERS": "3", "NTL_DEVICE_STREAMS": "3"}, 256_000_000)]
        for env, batch in configs:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--run", d, "--batch", str(batch)], env=dict(os.environ, **env), check=False)
        if a.forms:
            # the reference's real input forms (ntLink:113-117,222): one .fa.gz, several .fq.gz -- on the first 2 files
            import gzip
            sub = files[:1]
            t0 = time.perf_counter()
            src = os.path.join(d, sub[0])
            gz1 = os.path.join(d, "one.fa.gz")
            subprocess.check_call(f"head -c 2000000000 {src} | gzip -1 > {gz1}", shell=True)
            print(json.dumps({"gz_prepared_s": round(time.perf_counter() - t0, 1), "gz_bytes": os.path.getsize(gz1)}), flush=True)
            for env in ({}, {"NTL_IO_THREADS": "64"}):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--run", d, "--batch", "256000000", "--files", "one.fa.gz"],
                               env=dict(os.environ, **env), check=False)
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()

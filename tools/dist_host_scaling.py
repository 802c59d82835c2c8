#!/usr/bin/env python3
"""What N ranks do to ONE host: `ntlink_amd.dist_pair` file to file at world 1 / 2 / 4 / 8 on one box -- every rank
on the box's single GPU (NTL_DIST_ONE_DEVICE=0), so what is measured is the HOST side of BASELINE configs[3]: N parsers, N writers, N
Python drivers and one packed-contig copy under the box's CPU quota (the GPU boxes of this project: 256 CPUs visible, 16 granted).  The
GPU is shared N ways here and not on an 8-GPU node; the device seconds each rank reports say how much of a run that is.
Inputs: the C3 assembly + `--bases` of C3 reads as 8 plain FASTA files, and the same reads as one BGZF FASTQ file, in /dev/shm.
usage: tools/dist_host_scaling.py [--bases 8e9] [--worlds 1,2,4,8] [-o profiles/r06_dist_host_scaling.json]"""
import argparse
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--bases", type=float, default=8e9)
ap.add_argument("--bgzf-bases", type=float, default=2e9)
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--workload", default="C3")
ap.add_argument("-o", "--out", default=None)
a = ap.parse_args()


def make_inputs(d):
    import bench
    from ntlink_amd import capi, synth
    dev = capi.Device(0)
    wl = synth.DeviceWorkload(dev, a.workload, with_reads=False)
    W = wl.W
    cbuf, coff = wl.contigs.download()
    bench.write_fasta(os.path.join(d, "asm.fa"), cbuf, coff, b"ctg")
    del cbuf
    files, total = [], 0
    nb = 8
    first = None
    for b in range(nb):
        rb, _ = wl.make_reads(int(a.bases) // nb, seed=(91, b))
        rbuf, roff = rb.download()
        rb.close()
        p = os.path.join(d, f"reads_{b:02d}.fa")
        bench.write_fasta(p, rbuf, roff, b"r%d_" % b)
        files.append(os.path.basename(p))
        total += int(roff[-1])
        if first is None:
            first = (rbuf, roff)
        else:
            del rbuf
    # the first files' reads again as ONE bgzip'd FASTQ file (zlib level 1, 64 blocks per task)
    rbuf, roff = first
    n = int(min(len(roff) - 1, max(1, (len(roff) - 1) * a.bgzf_bases * nb / a.bases)))
    parts = []
    for i in range(n):
        s = bytes(rbuf[int(roff[i]):int(roff[i + 1])])
        parts.append(b"@r0_%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")
    whole = b"".join(parts)
    bgzf_bases = int(roff[n])
    del parts, rbuf
    blocks = [whole[i:i + 0xFF00] for i in range(0, len(whole), 0xFF00)] + [b""]

    def bgzf_blocks(lo):
        out = []
        for ch in blocks[lo:lo + 64]:
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = co.compress(ch) + co.flush()
            out.append(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1)
                       + body + struct.pack("<II", zlib.crc32(ch) & 0xFFFFFFFF, len(ch)))
        return b"".join(out)

    with ThreadPoolExecutor(32) as ex, open(os.path.join(d, "all.fq.bgz.gz"), "wb") as fh:
        for piece in ex.map(bgzf_blocks, range(0, len(blocks), 64)):
            fh.write(piece)
    del whole, blocks
    wl.close()
    dev.close()
    return W, files, total, bgzf_bases


def one(d, world, reads, prefix, W, port):
    for f in os.listdir(d):
        if f.startswith(prefix + ".") or f.startswith("asm.fa.k"):
            os.remove(os.path.join(d, f))
    env = dict(os.environ, NTL_DIST_ONE_DEVICE="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port",
           str(port), "-m", "ntlink_amd.dist_pair", "pair", "target=asm.fa", f"reads={reads}", f"prefix={prefix}", f"k={W['k']}", f"w={W['w']}", "paf=True",
           "ntlink_pairs_tsv=True", "v=1"]
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    t0 = time.perf_counter()
    p = subprocess.run(cmd, cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1800)
    wall = time.perf_counter() - t0
    if p.returncode != 0:
        return {"world": world, "error": p.stdout[-1500:]}
    rep = {}
    tf = os.path.join(d, f"{prefix}.n1.scaffold.dot.time")
    for line in open(tf):
        if ": " in line:
            key, val = line.strip().split(": ", 1)
            rep[key] = val
    return {"world": world, "wall_s_incl_process_start": round(wall, 2), "report": rep}


def main():
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(prefix="ntl_dist_", dir=base)
    out = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "dir": d}
    try:
        t0 = time.perf_counter()
        W, files, total, bgzf_bases = make_inputs(d)
        out["prepare_inputs_s"] = round(time.perf_counter() - t0, 1)
        out["read_bases_fasta"], out["read_bases_bgzf"] = total, bgzf_bases
        import bench
        out["host_cpu"] = dict(zip(("cpus_visible", "cpu_quota_cores"), bench.cpu_budget()))
        rows = []
        port = 29700
        for form, reads, bases in (("plain_fasta_8_files", " ".join(files), total), ("one_bgzf_fq_gz", "all.fq.bgz.gz", bgzf_bases)):
            for world in [int(x) for x in a.worlds.split(",")]:
                port += 1
                r = one(d, world, reads, "run", W, port)
                r["input"] = form
                rep = r.get("report", {})
                el = None
                for key in ("Elapsed (wall clock) time (h:mm:ss or m:ss)", "Elapsed (wall clock) seconds"):
                    if key in rep:
                        v = rep[key]
                        try:
                            el = float(v) if ":" not in v else sum(float(x) * 60 ** i for i, x in enumerate(reversed(v.split(":"))))
                        except ValueError:
                            pass
                if el:
                    r["elapsed_s"] = el
                    r["aggregate_Gbases_per_s"] = round(bases / el / 1e9, 2)
                r["aggregate_Gbases_per_s_incl_process_start"] = round(bases / r["wall_s_incl_process_start"] / 1e9, 2) if "wall_s_incl_process_start" in r else None
                rows.append(r)
                print(json.dumps({k: v for k, v in r.items() if k != "report"}), flush=True)
        out["runs"] = rows
    finally:
        shutil.rmtree(d, ignore_errors=True)
    txt = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    else:
        print(txt)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""End-to-end (file to file) timing of `ntLink pair` on synthetic data: FASTA in the page cache ->
.tsv/.verbose_mapping.tsv/.paf/.pairs.tsv/.dot on disk.  Complements bench.py (device-resident).
Usage: tools/e2e_bench.py [--scale S] [--gz]"""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ntlink_amd import capi, pipeline, synth  # noqa: E402


FASTQ = False  # --fastq: four-line records with a constant quality string


def write_fasta(path, buf, off, names, gz=False):
    opener = (lambda p: gzip.open(p, "wb", compresslevel=1)) if gz else (lambda p: open(p, "wb"))
    with opener(path) as f:
        raw = buf.tobytes()
        for i, n in enumerate(names):
            s = raw[int(off[i]):int(off[i + 1])]
            if FASTQ:
                f.write(b"@" + n.encode() + b" some description\n" + s + b"\n+\n" + b"@" * len(s) + b"\n")  # '@' qualities: the hard case
            else:
                f.write(b">" + n.encode() + b"\n" + s + b"\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.2)
    ap.add_argument("--workload", default="C2", help="C2, or C3 / C5 with --read-bases (their full read sets are 90 / 180 Gbases)")
    ap.add_argument("--read-bases", type=int, default=0, help="override the workload's read bases")
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--batch", type=int, default=pipeline.DEFAULT_BATCH_BASES)
    ap.add_argument("--files", type=int, default=1, help="split the reads over this many files (FASTQ when --gz)")
    ap.add_argument("--fastq", action="store_true", help="reads as four-line FASTQ (qualities all '@')")
    ap.add_argument("--pipe", action="store_true", help="time the reference's two-operator recipe (bin/indexlr | bin/ntlink_pair.py) instead")
    ap.add_argument("--stages", action="store_true", help="also time every stage of one whole-input batch on its own")
    a = ap.parse_args()
    global FASTQ
    W = synth.workload(a.workload, a.scale)
    if a.read_bases:
        W["read_bases"] = a.read_bases
    chroms, cbuf, coff, cn, _ = synth.make_assembly(1, W["n_chrom"], W["contigs_per_chrom"], W["contig_len"])
    rbuf, roff, rn = synth.make_reads(2, chroms, W["read_bases"], W["read_len"], W["sub"], W["ins"], W["dele"], lognormal_sigma=0.4)
    d = tempfile.mkdtemp(prefix="ntl_e2e_")
    tgt, rds = os.path.join(d, "asm.fa"), os.path.join(d, "reads.fa" + (".gz" if a.gz else ""))
    write_fasta(tgt, cbuf, coff, cn)
    FASTQ = a.fastq
    if a.files == 1:
        write_fasta(rds, rbuf, roff, rn, a.gz)
        read_arg = os.path.basename(rds)
    else:
        from concurrent.futures import ThreadPoolExecutor
        per = (len(rn) + a.files - 1) // a.files
        parts = []

        def one(f):
            lo, hi = f * per, min(len(rn), (f + 1) * per)
            path = os.path.join(d, f"reads_{f:03d}.fa" + (".gz" if a.gz else ""))
            write_fasta(path, rbuf[int(roff[lo]):int(roff[hi])], roff[lo:hi + 1] - roff[lo], rn[lo:hi], a.gz)
            return os.path.basename(path)

        with ThreadPoolExecutor(16) as ex:
            parts = list(ex.map(one, range(a.files)))
        read_arg = " ".join(parts)
    if a.pipe:  # ntLink:198-199,221-225 with the drop-in executables, three processes and a pipe
        import subprocess
        env = dict(os.environ, PATH=os.path.join(ROOT, "bin") + os.pathsep + os.environ["PATH"])
        k, w = W["k"], W["w"]
        cat = "gzip -cd -f" if a.gz else "cat"
        sh = (f"indexlr --long --pos --strand -k {k} -w {w} -t 8 asm.fa > asm.fa.k{k}.w{w}.tsv && "
              f"{cat} {read_arg} | indexlr --long --pos --strand --len -k {k} -w {w} -t 8 - | "
              f"ntlink_pair.py -p out -n 1 -m asm.fa.k{k}.w{w}.tsv -s asm.fa -k {k} -a 1 -z 1000 -f 10 -x 0 --verbose --pairs --paf -")
        t0 = time.perf_counter()
        subprocess.check_call(["bash", "-e", "-o", "pipefail", "-c", sh], cwd=d, env=env, stdout=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        nb = int(roff[-1])
        # the fused driver on the same files must leave byte-identical outputs (different batch cuts, no text between)
        import hashlib
        dev = capi.Device(0)
        os.chdir(d)
        pipeline.run_pair(dev, "asm.fa", read_arg, k=k, w=w, paf=True, pairs_tsv=True, batch_bases=a.batch, prefix="fused", write_contig_tsv=False)
        dev.close()
        same = {}
        for ext in (".verbose_mapping.tsv", ".paf", ".pairs.tsv", ".n1.scaffold.dot"):
            h = [hashlib.md5(open(pre + ext, "rb").read()).hexdigest() for pre in ("out", "fused")]
            same[ext] = h[0] == h[1]
        print(json.dumps({"mode": "indexlr | ntlink_pair.py (three processes, text between them)", "seconds": round(dt, 3),
                          "end_to_end_Gbases_per_s": round(nb / dt / 1e9, 4), "read_bases": nb, "gz": a.gz,
                          "outputs_equal_fused_driver": same}))
        if not all(same.values()):
            sys.exit(1)
        return
    dev = capi.Device(0)
    os.chdir(d)
    t0 = time.perf_counter()
    st = pipeline.run_pair(dev, "asm.fa", read_arg, k=W["k"], w=W["w"], paf=True, pairs_tsv=True, batch_bases=a.batch,
                           sensitive=W["sensitive"])
    dt = time.perf_counter() - t0
    out_bytes = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.startswith("asm.fa."))
    import hashlib
    md5 = {ext: hashlib.md5(open(os.path.join(d, f"asm.fa.k{W['k']}.w{W['w']}.z1000" + ext), "rb").read()).hexdigest()[:12]
           for ext in (".verbose_mapping.tsv", ".paf", ".pairs.tsv")}
    print(json.dumps({"end_to_end_Gbases_per_s": round(st["read_bases"] / dt / 1e9, 4), "seconds": round(dt, 3),
                      "workload": a.workload, "assembly_bp": int(coff[-1]), "read_bases": st["read_bases"], "reads": st["reads"], "gz": a.gz, "fastq": a.fastq, "files": a.files, "batch_bases": a.batch, "output_bytes": out_bytes, "md5": md5,
                      "t_contig_stage": round(st["t_contigs"], 3), "t_wait_for_ingest": round(st["t_ingest"], 3), "t_device_incl_pack_pcie": round(st["t_device"], 3),
                      "t_handover": round(st["t_handover"], 3), "t_drain_tail": round(st.get("t_drain_tail", 0), 3), "t_graph": round(st.get("t_graph", 0), 3),
                      "t_write": round(st["t_write"], 3), "t_tally": round(st["t_tally"], 3), "device": dev.name}))
    if a.stages:
        from ntlink_amd import formats, seqio
        T = {}

        def lap(name, t0):
            dev.sync()
            T[name] = round(time.perf_counter() - t0, 4)

        t = time.perf_counter(); ctg = seqio.load_all(["asm.fa"]); rs = seqio.load_all([os.path.basename(rds)], alloc=dev.pinned_empty); lap("parse_first", t)
        dev.pinned_release(rs.buf)
        t = time.perf_counter(); rs = seqio.load_all([os.path.basename(rds)], alloc=dev.pinned_empty); lap("parse_reused_buffer", t)
        t = time.perf_counter(); cb = dev.batch(ctg.buf, ctg.offsets); csk = dev.sketch(cb, W["k"], W["w"]); ix = dev.index(csk, ctg.lengths); lap("contigs", t)
        t = time.perf_counter(); rb = dev.batch(rs.buf, rs.offsets); lap("pack_h2d", t)
        t = time.perf_counter(); rsk = dev.sketch(rb, W["k"], W["w"]); lap("sketch", t)
        t = time.perf_counter(); res = dev.map(ix, rsk, rs.lengths, k=W["k"]); lap("map", t)
        t = time.perf_counter(); rec = res.download(); lap("download", t)
        t = time.perf_counter()
        with open("stage.verbose", "w") as fh:
            formats.write_verbose(fh, rec, rs.names, ctg.names)
        with open("stage.paf", "w") as fh:
            formats.write_paf(fh, rec, rs.names, rs.lengths, ctg.names, ctg.lengths)
        lap("write", t)
        for h in (res, rsk, rb, ix, csk, cb):
            h.close()
        print(json.dumps({"stages_s": T, "io_threads": os.environ.get("NTL_IO_THREADS", "default"), "cores": os.cpu_count()}))
    dev.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""the overlap stage's sketch (`indexlr --long --pos -k 15 -w 5`, ntLink:243-251) on the
device: one minimizer per three bases, so the emit path and the TSV writer carry the load.
Prints one JSON line: device-resident sketch rate, records/s, and the `--pos`-only TSV emit rate."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ntlink_amd import capi, formats, synth  # noqa: E402


def main():
    k, w = 15, 5
    W = synth.workload("C2", 4.0)
    _chroms, cbuf, coff, cn, _ = synth.make_assembly(1, W["n_chrom"], W["contigs_per_chrom"], W["contig_len"])
    dev = capi.Device(0)
    with dev.batch(cbuf, coff) as b:
        with dev.sketch(b, k, w) as sk:  # warm-up
            n = sk.count
        dev.sync()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            with dev.sketch(b, k, w) as sk:
                pass
        dev.sync()
        dt = (time.perf_counter() - t0) / reps
        with dev.sketch(b, k, w) as sk:
            t1 = time.perf_counter(); off, h, p, s = sk.download(); t_dl = time.perf_counter() - t1
    import numpy as np
    lens = np.diff(coff).astype(np.uint32)
    with tempfile.NamedTemporaryFile("w", suffix=".tsv") as fh:
        t2 = time.perf_counter()
        formats.write_indexlr(fh, cn, lens, off, h, p, s, False, with_strand=False)
        fh.flush()
        t_wr = time.perf_counter() - t2
        nbytes = os.path.getsize(fh.name)
    bases = int(coff[-1])
    print(json.dumps({"workload": f"{bases} bp assembly, k={k} w={w} (overlap stage, ntLink:243-251)", "minimizers": int(n),
                      "density": round(n / bases, 4), "sketch_ms": round(dt * 1e3, 3), "sketch_Gbases_per_s": round(bases / dt / 1e9, 1),
                      "records_G_per_s": round(n / dt / 1e9, 2), "download_s": round(t_dl, 3), "tsv_bytes": nbytes,
                      "tsv_write_s": round(t_wr, 3), "tsv_GB_per_s": round(nbytes / t_wr / 1e9, 2), "device": dev.name}))
    dev.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden vectors for ntlink_amd.overlap: the reference's own read_minimizers / read_minimizers_path
(bin/ntlink_overlap_sequences.py:145-190), imported in the build container (`igraph` stubbed: the module only uses it for the
graph), on `indexlr --long --pos -k K -w W` TSVs written from the oracle's sketch (the reference's TSV producer is btllib's
indexlr, absent here; the oracle's sketch is pinned by the reference's goldens) with seeded valid regions.
Inputs: the four fixture assemblies at k15 w5 (the overlap stage's parameters, ntLink:46-47) plus a synthetic assembly with
tandem repeats, homopolymers and copied segments, so that duplicated minimizers inside a contig and inside / outside the
valid regions all occur.  Output under tests/golden/gen/overlap/: <case>.tsv.gz (input), <case>.json.gz (valid regions, expected
mxs and positions per contig, and for the path form the expected split at the LAST markers).  Only data is stored."""
import gzip
import json
import os
import random
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.modules["igraph"] = types.ModuleType("igraph")
sys.path.insert(0, "/root/reference/bin")
sys.path.insert(0, REPO)
import ntlink_overlap_sequences as ref  # noqa: E402  (the reference)
import oracle  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden", "gen", "overlap")
GOLD_REF = os.path.join(REPO, "tests", "golden", "ref")


def synthetic(rng):
    def rnd(n):
        return "".join(rng.choice("ACGT") for _ in range(n))
    unit, seg = rnd(23), rnd(700)
    return [("tandem", rnd(900) + unit * 30 + rnd(900)), ("copies", rnd(500) + seg + rnd(300) + seg + rnd(400) + seg[:350]),
            ("homopoly", rnd(300) + "A" * 200 + rnd(300) + "ACAC" * 60 + rnd(100)), ("plain", rnd(4000)),
            ("withN", rnd(600) + "N" * 50 + rnd(600) + "n" + rnd(300)), ("tiny", rnd(18)), ("empty_sketch", "ACG")]


def write_tsv(path, records, k, w):
    buf = np.frombuffer("".join(s for _, s in records).encode(), np.uint8)
    off = np.zeros(len(records) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for _, s in records])
    mx_off, h, p, _s = oracle.sketch_batch(buf, off, k, w)
    with gzip.open(path, "wt") as f:
        for i, (name, _) in enumerate(records):
            a, b = int(mx_off[i]), int(mx_off[i + 1])
            # indexlr prints the id and a tab even when the record has no minimizers
            f.write(name + "\t" + " ".join(f"{h[j]}:{p[j]}" for j in range(a, b)) + "\n")


def regions(records, rng):
    valid = {}
    for name, s in records:
        r = rng.random()
        n = len(s)
        if r < 0.15:
            continue  # contig not in valid_mx_positions at all
        if r < 0.25:
            valid[name] = []
        elif r < 0.55:
            valid[name] = [(0, rng.randrange(0, n + 50))]
        elif r < 0.8:
            a = rng.randrange(0, max(1, n))
            valid[name] = [(a, n + 10)]
        else:
            a, b = sorted((rng.randrange(0, n + 1), rng.randrange(0, n + 1)))
            c, d = sorted((rng.randrange(0, n + 1), rng.randrange(0, n + 1)))
            valid[name] = [(a, b), (c, d)]  # may overlap
    return valid


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = []
    inputs = [("scaffolds_%d" % i, list(oracle.read_fastx(os.path.join(GOLD_REF, "scaffolds_%d.fa" % i)))) for i in (1, 2, 3, 4)]
    inputs = [(n, [(a, s.decode() if isinstance(s, bytes) else s) for a, s in recs]) for n, recs in inputs]
    inputs.append(("synthetic", synthetic(random.Random(99))))
    for name, recs in inputs:
        for k, w, seed in ((15, 5, 1), (15, 5, 2)) if name != "synthetic" else ((15, 5, 1), (15, 5, 2), (8, 3, 3), (20, 10, 4)):
            if name in ("scaffolds_2", "scaffolds_3") and seed == 2:
                continue
            case = "%s_k%d_w%d_s%d" % (name, k, w, seed)
            tsv = os.path.join(OUT, "%s_k%d_w%d.tsv.gz" % (name, k, w))
            if not os.path.exists(tsv):
                write_tsv(tsv, recs, k, w)
            valid = regions(recs, random.Random(seed * 7919 + len(name)))
            with tempfile.NamedTemporaryFile("w", suffix=".tsv") as tmp:
                tmp.write(gzip.open(tsv, "rt").read())
                tmp.flush()
                mx_info, mxs = ref.read_minimizers(tmp.name, valid)
                # path form: LAST markers after every third record
                text = gzip.open(tsv, "rt").read().splitlines(keepends=True)
                with_markers, chunks = [], []
                for i, line in enumerate(text):
                    with_markers.append(line)
                    if i % 3 == 2:
                        with_markers.append("LASTntLink_%d\t\n" % (i // 3))
                reader = iter(with_markers)
                while True:
                    info_p, mxs_p = ref.read_minimizers_path(reader, valid)
                    chunks.append({n: m[0] for n, m in mxs_p.items()})
                    try:
                        nxt = next(reader)
                    except StopIteration:
                        break
                    reader = iter([nxt] + list(reader))
            for n in mxs:
                assert list(mx_info[n]) == mxs[n][0] and len(mxs[n]) == 1
            exp = {n: {"mxs": mxs[n][0], "pos": [mx_info[n][m][1] for m in mxs[n][0]]} for n in mxs}
            json.dump({"k": k, "w": w, "tsv": os.path.basename(tsv), "valid": valid, "expected": exp, "path_chunks": chunks},
                      gzip.open(os.path.join(OUT, case + ".json.gz"), "wt"))
            cases.append(case)
            print(case, "contigs", len(exp), "kept", sum(len(v["mxs"]) for v in exp.values()))
    json.dump(cases, open(os.path.join(OUT, "cases.json"), "w"))


if __name__ == "__main__":
    main()

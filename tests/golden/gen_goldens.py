#!/usr/bin/env python3
"""Generate golden vectors for the mapping half of the path by IMPORTING the reference's Python.

Runs only in the build container (needs /root/reference); nothing under tests/ or the product
imports this script.  What it does:

1. stubs ``igraph`` (absent here; the reference touches it only in the graph tail,
   bin/ntlink_pair.py:267,294-303,502-505) and imports ``ntlink_pair`` / ``ntlink_utils`` /
   ``ntlink_paf_output`` from /root/reference/bin with bytecode writing disabled;
2. for every fixture pair the reference's tests use, sketches contigs and reads with the oracle
   (btllib is not installed; the oracle sketch is pinned by the reference's contig TSV goldens),
   feeds the TSV text through the reference's own ``read_minimizers`` / ``find_scaffold_pairs`` /
   filters / ``write_pairs`` and stores ``.verbose_mapping.tsv``, ``.paf``, ``.pairs.tsv``;
3. does the same for seeded synthetic scenarios built to hit the branches the fixtures never
   take (--sensitive, --repeat-filter, x != 0, z filter, noisy contigs, subsumption, every
   PAF filter/break branch, duplicate contig positions, > f contigs in a read).

Outputs: tests/golden/gen/ (data only: TSV inputs, parameters, expected outputs).
"""
import argparse
import gzip
import hashlib
import json
import os
import random
import shutil
import sys
import tempfile
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.modules["igraph"] = types.ModuleType("igraph")
sys.path.insert(0, "/root/reference/bin")

import ntlink_pair  # noqa: E402  (the reference)
import ntlink_utils  # noqa: E402

import oracle  # noqa: E402

REF = os.path.join(REPO, "tests", "golden", "ref")
OUT = os.path.join(REPO, "tests", "golden", "gen")


def run_reference(contig_tsv, reads_tsv, target_fa, prefix, k, z=1000, a=1, f=10, x=0.0, n=1,
                  sensitive=False, repeat_filter=False):
    """Drive the reference exactly as NtLink.main does (bin/ntlink_pair.py:560-596), minus igraph."""
    nt = object.__new__(ntlink_pair.NtLink)
    nt.args = argparse.Namespace(FILES=[reads_tsv], s=target_fa, m=contig_tsv, p=prefix, n=n, k=k, z=z, a=a,
                                 f=f, x=x, checkpoint=None, pairs=True, paf=True, sensitive=sensitive,
                                 repeat_filter=repeat_filter, verbose=True)
    with ntlink_utils.HiddenPrints():
        ntlink_pair.NtLink.list_mx_info = nt.read_minimizers()
        ntlink_pair.NtLink.scaffolds = ntlink_utils.read_fasta_file(target_fa)
        pairs = nt.find_scaffold_pairs()
        pairs = nt.filter_pairs_distances(pairs)
        pairs = nt.filter_weak_anchor_pairs(pairs)
        nt.write_pairs(pairs)


def sketch_file(path, k, w, with_len):
    recs = []
    for name, seq in oracle.read_fastx(path):
        h, p, s = oracle.sketch_seq(seq, k, w)
        recs.append((name, len(seq), h, p, s))
    return oracle.format_indexlr(recs, with_len=with_len)


FIXTURES = [
    # (name, target, reads, k, w)
    ("t1_k32_w250", "scaffolds_1.fa", "long_reads_1.fa", 32, 250),
    ("t2_k32_w100", "scaffolds_2.fa", "long_reads_2.fq.gz", 32, 100),
    ("t3_k24_w250", "scaffolds_3.fa", "long_reads_3.fa.gz", 24, 250),
    ("t4_k40_w100", "scaffolds_4.fa", "long_reads_4.fa.gz", 40, 100),
    ("t7_top5_k40_w100", "scaffolds_4.fa", "long_reads_4_top5.fa", 40, 100),
    ("c1_k32_w100", "scaffolds_1.fa", "long_reads_1.fa", 32, 100),
]


def gen_fixtures(tmp):
    summary = {}
    d = os.path.join(OUT, "fixtures")
    os.makedirs(d, exist_ok=True)
    for name, target, reads, k, w in FIXTURES:
        ctsv = os.path.join(tmp, name + ".contigs.tsv")
        rtsv = os.path.join(tmp, name + ".reads.tsv")
        ctext = sketch_file(os.path.join(REF, target), k, w, False)
        rtext = sketch_file(os.path.join(REF, reads), k, w, True)
        open(ctsv, "w").write(ctext)
        open(rtsv, "w").write(rtext)
        prefix = os.path.join(tmp, name)
        for flags in ({}, {"sensitive": True}, {"repeat_filter": True}):
            tag = name + "".join("." + f for f in flags)
            for ext in (".verbose_mapping.tsv", ".paf", ".pairs.tsv"):
                if os.path.exists(prefix + ext):
                    os.remove(prefix + ext)
            run_reference(ctsv, rtsv, os.path.join(REF, target), prefix, k, **flags)
            for ext in (".verbose_mapping.tsv", ".paf", ".pairs.tsv"):
                shutil.copy(prefix + ext, os.path.join(d, tag + ext))
            summary[tag] = {
                "target": target, "reads": reads, "k": k, "w": w, **flags,
                "contig_tsv_md5": hashlib.md5(ctext.encode()).hexdigest(),
                "read_tsv_md5": hashlib.md5(rtext.encode()).hexdigest(),
                "read_minimizers": sum(len(l.split("\t")[2].split(" ")) for l in rtext.splitlines()
                                       if len(l.split("\t")) > 2),
            }
    json.dump(summary, open(os.path.join(d, "summary.json"), "w"), indent=1, sort_keys=True)


# ------------------------------------------------------------------ synthetic scenarios

def make_scenario(seed, n_ctg, n_reads, k, dense):
    """A random assembly index + reads, as indexlr-format TSV text."""
    rng = random.Random(seed)
    ctg_len = [rng.choice([400, 800, 999, 1000, 1001, 3000, 12000, 40000]) for _ in range(n_ctg)]
    ctg_len[0] = 40000
    ctg_len[1] = 25000
    names = [f"ctg{j:03d}" if seed % 2 else f"s{(j * 7919) % 1000}_{j}" for j in range(n_ctg)]
    used = set()

    def new_key():
        while True:
            v = rng.getrandbits(64)
            if v not in used:
                used.add(v)
                return v

    ctg_mx = []  # per contig [(key, pos, strand)]
    step = 40 if dense else 180
    for j in range(n_ctg):
        pos, lst = rng.randrange(0, 50), []
        while pos + k <= ctg_len[j]:
            lst.append((new_key(), pos, rng.choice("+-")))
            pos += rng.randrange(1, 2 * step)
        ctg_mx.append(lst)
    # keys duplicated inside / across contigs: must vanish from the index
    flat = [(j, i) for j in range(n_ctg) for i in range(len(ctg_mx[j]))]
    for _ in range(max(2, len(flat) // 25)):
        (j1, i1), (j2, i2) = rng.sample(flat, 2)
        key = ctg_mx[j1][i1][0]
        ctg_mx[j2][i2] = (key, ctg_mx[j2][i2][1], ctg_mx[j2][i2][2])
    contig_lines = []
    for j in range(n_ctg):
        if ctg_mx[j]:
            contig_lines.append(names[j] + "\t" + " ".join(f"{a}:{b}:{c}" for a, b, c in ctg_mx[j]))
        else:
            contig_lines.append(names[j])
    read_lines = []
    for r in range(n_reads):
        toks = []  # (key, strand) in read order
        nseg = rng.choice([1, 1, 2, 2, 3, 4, 6, 12])
        for _ in range(nseg):
            j = rng.randrange(n_ctg)
            if not ctg_mx[j]:
                continue
            lo = rng.randrange(len(ctg_mx[j]))
            hi = min(len(ctg_mx[j]), lo + rng.choice([1, 2, 3, 5, 9, 20, 40]))
            seg = ctg_mx[j][lo:hi]
            flip = rng.random() < 0.5
            if flip:
                seg = seg[::-1]
            seg = [(key, (st if not flip else ("-" if st == "+" else "+"))) for key, _p, st in seg]
            seg = [t if rng.random() > 0.1 else (t[0], rng.choice("+-")) for t in seg]
            seg = [t for t in seg if rng.random() > 0.15]
            mode = rng.random()
            if mode < 0.25 and len(seg) > 2:      # local disorder: swap neighbours
                for _s in range(rng.choice([1, 1, 2, 4])):
                    i = rng.randrange(len(seg) - 1)
                    seg[i], seg[i + 1] = seg[i + 1], seg[i]
            elif mode < 0.35 and len(seg) > 3:    # move one element far away
                i = rng.randrange(len(seg))
                e = seg.pop(i)
                seg.insert(rng.randrange(len(seg) + 1), e)
            elif mode < 0.42 and len(seg) > 4:    # block transposition
                c = rng.randrange(1, len(seg))
                seg = seg[c:] + seg[:c]
            elif mode < 0.47:                     # shuffle
                rng.shuffle(seg)
            if rng.random() < 0.2 and seg:        # the same contig minimizer hit twice
                i = rng.randrange(len(seg))
                seg.insert(rng.randrange(len(seg) + 1), seg[i])
            toks.extend(seg)
            if rng.random() < 0.3:                # stray hit on another contig
                j2 = rng.randrange(n_ctg)
                if ctg_mx[j2]:
                    key, _p, st = rng.choice(ctg_mx[j2])
                    toks.insert(rng.randrange(len(toks) + 1), (key, st))
        for _ in range(rng.choice([0, 1, 3, 10])):  # minimizers that are not in the assembly
            toks.insert(rng.randrange(len(toks) + 1), (new_key(), rng.choice("+-")))
        pos, out = rng.randrange(0, 300), []
        for key, st in toks:
            out.append(f"{key}:{pos}:{st}")
            pos += rng.randrange(1, 2 * step)
        rlen = pos + k + rng.randrange(0, 200)
        if rng.random() < 0.15:
            rlen = max(pos + k, rlen // 1)  # keep valid: every k-mer must fit in the read
        name = f"read{r}"
        if out:
            read_lines.append(f"{name}\t{rlen}\t" + " ".join(out))
        else:
            read_lines.append(f"{name}\t{rlen}")
    return names, ctg_len, "\n".join(contig_lines) + "\n", "\n".join(read_lines) + "\n"


SCENARIOS = [
    # (name, seed, n_ctg, n_reads, k, dense, params)
    ("syn_default", 11, 6, 400, 32, False, {}),
    ("syn_dense", 12, 5, 300, 24, True, {}),
    ("syn_sensitive", 13, 7, 400, 32, False, {"sensitive": True}),
    ("syn_repeat", 14, 6, 400, 32, False, {"repeat_filter": True}),
    ("syn_sens_repeat", 15, 8, 300, 40, True, {"sensitive": True, "repeat_filter": True}),
    ("syn_x15", 16, 6, 400, 32, False, {"x": 1.5}),
    ("syn_x03", 17, 6, 300, 32, True, {"x": 0.3}),
    ("syn_z500_f3", 18, 12, 400, 32, False, {"z": 500, "f": 3}),
    ("syn_a2", 19, 5, 400, 32, False, {"a": 2}),
    ("syn_many_ctg", 20, 40, 300, 20, False, {"f": 10}),
]


def _gz_write(path, text):
    with open(path, "wb") as raw, gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0) as f:
        f.write(text.encode())


def gen_scenarios(tmp):
    d = os.path.join(OUT, "synthetic")
    os.makedirs(d, exist_ok=True)
    for name, seed, n_ctg, n_reads, k, dense, params in SCENARIOS:
        names, ctg_len, ctext, rtext = make_scenario(seed, n_ctg, n_reads, k, dense)
        ctsv, rtsv = os.path.join(tmp, name + ".c.tsv"), os.path.join(tmp, name + ".r.tsv")
        fa = os.path.join(tmp, name + ".fa")
        open(ctsv, "w").write(ctext)
        open(rtsv, "w").write(rtext)
        with open(fa, "w") as f:
            for nm, ln in zip(names, ctg_len):
                f.write(f">{nm} comment\n{'A' * ln}\n")
        prefix = os.path.join(tmp, name)
        run_reference(ctsv, rtsv, fa, prefix, k, **params)
        _gz_write(os.path.join(d, name + ".contigs.tsv.gz"), ctext)
        _gz_write(os.path.join(d, name + ".reads.tsv.gz"), rtext)
        json.dump({"k": k, "params": params, "ctg_names": names, "ctg_len": ctg_len},
                  open(os.path.join(d, name + ".json"), "w"))
        for ext in (".verbose_mapping.tsv", ".paf", ".pairs.tsv"):
            _gz_write(os.path.join(d, name + ext + ".gz"), open(prefix + ext).read())
        nv = sum(1 for _ in open(prefix + ".verbose_mapping.tsv"))
        npaf = sum(1 for _ in open(prefix + ".paf"))
        npairs = sum(1 for _ in open(prefix + ".pairs.tsv"))
        print(f"{name}: verbose={nv} paf={npaf} pairs={npairs}")


def gen_checkpoint(tmp):
    """Checkpoint mode (bin/ntlink_pair.py:437-488): pairs re-derived from a verbose_mapping.tsv."""
    d = os.path.join(OUT, "fixtures")
    for tag, target, k in (("t4_k40_w100", "scaffolds_4.fa", 40), ("t3_k24_w250", "scaffolds_3.fa", 24)):
        prefix = os.path.join(tmp, "ck_" + tag)
        shutil.copy(os.path.join(d, tag + ".verbose_mapping.tsv"), prefix + ".verbose_mapping.tsv")
        nt = object.__new__(ntlink_pair.NtLink)
        nt.args = argparse.Namespace(FILES=["-"], s=os.path.join(REF, target), m=None, p=prefix, n=1, k=k, z=1000, a=1,
                                     f=10, x=0.0, checkpoint=prefix + ".verbose_mapping.tsv", pairs=True, paf=False,
                                     sensitive=False, repeat_filter=False, verbose=True)
        with ntlink_utils.HiddenPrints():
            ntlink_pair.NtLink.list_mx_info = {}
            ntlink_pair.NtLink.scaffolds = ntlink_utils.read_fasta_file(nt.args.s)
            pairs = nt.find_scaffold_pairs_checkpoints()
            pairs = nt.filter_pairs_distances(pairs)
            pairs = nt.filter_weak_anchor_pairs(pairs)
            nt.write_pairs(pairs)
        shutil.copy(prefix + ".pairs.tsv", os.path.join(d, tag + ".checkpoint.pairs.tsv"))


if __name__ == "__main__":
    tmp = tempfile.mkdtemp(prefix="ntl_gold_")
    try:
        gen_fixtures(tmp)
        gen_scenarios(tmp)
        gen_checkpoint(tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

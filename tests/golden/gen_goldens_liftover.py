#!/usr/bin/env python3
"""Golden vectors for ntlink_amd.liftover: the reference's own liftover
(bin/ntlink_liftover_mappings.py:39-129: read_agp, liftover_mappings), imported in the build container, on
  (a) the four `.verbose_mapping.tsv` + `.trimmed_scafs.agp` pairs the reference ships under tests/expected_outputs
      (both copied as DATA to tests/golden/ref/expected_outputs/), and
  (b) the verbose mappings of the synthetic scenarios through seeded random AGPs written here, built to reach what (a)
      never reaches: `-` components, partial component ranges (mappings outside are dropped), contigs missing from the
      AGP, unscaffolded contigs (path id == contig id), several contigs of a read in one path (concatenation,
      subsumption on path ids, the monotonicity filter).
Output: tests/golden/gen/liftover/<case>.agp (synthetic input), <case>.liftover.tsv.gz (expected output), cases.json.
Same import recipe as tests/golden/gen_goldens.py; only data is stored."""
import gzip
import json
import os
import random
import shutil
import sys
import tempfile
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.modules["igraph"] = types.ModuleType("igraph")
sys.path.insert(0, "/root/reference/bin")
import ntlink_liftover_mappings as ref  # noqa: E402  (the reference)

REF_EXP = "/root/reference/tests/expected_outputs"
GOLD_REF = os.path.join(REPO, "tests", "golden", "ref", "expected_outputs")
SYN = os.path.join(REPO, "tests", "golden", "gen", "synthetic")
OUT = os.path.join(REPO, "tests", "golden", "gen", "liftover")

FIXTURES = [("scaffolds_1.fa.k32.w250.z1000", 32), ("scaffolds_2.fa.k32.w100.z1000", 32),
            ("scaffolds_3.fa.k24.w250.z1000", 24), ("scaffolds_4.fa.k40.w100.z1000", 40)]
SCENARIOS = ["syn_default", "syn_sensitive", "syn_many_ctg", "syn_dense", "syn_repeat", "syn_x03"]


def random_agp(names, lens, seed, k):
    """AGP 2.0 lines: paths of 1-4 contigs (mostly neighbours in the list, so reads that span junctions land in one
    path), `N` gap lines between components, random orientation and component ranges."""
    rng = random.Random(seed)
    idx = list(range(len(names)))
    if rng.random() < 0.5:
        # swap a few neighbours: reads then meet the path's contigs in the wrong order
        for _ in range(max(1, len(idx) // 6)):
            a = rng.randrange(len(idx) - 1)
            idx[a], idx[a + 1] = idx[a + 1], idx[a]
    lines, path_no, i = [], 0, 0
    while i < len(idx):
        n = rng.choice([1, 1, 2, 3, 4])
        group = idx[i:i + n]
        i += n
        if rng.random() < 0.12:
            continue  # these contigs are not in the AGP at all
        if len(group) == 1 and rng.random() < 0.5:
            c = group[0]  # unscaffolded: the path is named after the contig
            lines.append("\t".join(map(str, (names[c], 1, lens[c], 1, "W", names[c], 1, lens[c], rng.choice("+-")))))
            continue
        pid = "ntLink_%d" % path_no
        path_no += 1
        at, comp = 1, 1
        for j, c in enumerate(group):
            lo, hi = 1, lens[c]
            if rng.random() < 0.5 and lens[c] > 4 * k:
                lo = 1 + rng.randrange(0, lens[c] // 4)
                hi = lens[c] - rng.randrange(0, lens[c] // 4)
            ori = rng.choice("++-")
            span = hi - lo + 1
            lines.append("\t".join(map(str, (pid, at, at + span - 1, comp, "W", names[c], lo, hi, ori))))
            at += span
            comp += 1
            if j + 1 < len(group):
                g = rng.choice([20, 100, 1337])
                kind = rng.choice(["N", "N", "P"])
                lines.append("\t".join(map(str, (pid, at, at + g - 1, comp, kind, g, "scaffold", "yes", "paired-ends"))))
                at += g
                comp += 1
    return "\n".join(lines) + "\n"


def run_ref(mappings_path, agp_path, k):
    with tempfile.NamedTemporaryFile("r", suffix=".tsv") as out:
        ref.liftover_mappings(mappings_path, ref.read_agp(agp_path), out.name, k)
        return open(out.name).read()


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = []
    for stem, k in FIXTURES:
        agp = stem + ".trimmed_scafs.agp"
        shutil.copyfile(os.path.join(REF_EXP, agp), os.path.join(GOLD_REF, agp))
        text = run_ref(os.path.join(GOLD_REF, stem + ".verbose_mapping.tsv"), os.path.join(GOLD_REF, agp), k)
        name = "fix_" + stem.split(".")[0]
        with gzip.open(os.path.join(OUT, name + ".liftover.tsv.gz"), "wt") as f:
            f.write(text)
        cases.append({"name": name, "k": k, "mappings": "ref/expected_outputs/" + stem + ".verbose_mapping.tsv",
                      "agp": "ref/expected_outputs/" + agp, "lines": text.count("\n")})
    for sc in SCENARIOS:
        meta = json.load(open(os.path.join(SYN, sc + ".json")))
        for seed in (1, 2, 3):
            name = "%s_agp%d" % (sc, seed)
            agp_text = random_agp(meta["ctg_names"], meta["ctg_len"], seed * 1000 + len(sc), meta["k"])
            agp_path = os.path.join(OUT, name + ".agp")
            open(agp_path, "w").write(agp_text)
            with tempfile.NamedTemporaryFile("w", suffix=".tsv") as m:
                m.write(gzip.open(os.path.join(SYN, sc + ".verbose_mapping.tsv.gz"), "rt").read())
                m.flush()
                text = run_ref(m.name, agp_path, meta["k"])
            with gzip.open(os.path.join(OUT, name + ".liftover.tsv.gz"), "wt") as f:
                f.write(text)
            cases.append({"name": name, "k": meta["k"], "mappings": "gen/synthetic/" + sc + ".verbose_mapping.tsv.gz",
                          "agp": "gen/liftover/" + name + ".agp", "lines": text.count("\n")})
    json.dump(cases, open(os.path.join(OUT, "cases.json"), "w"), indent=1)
    for c in cases:
        print(c["name"], c["lines"])


if __name__ == "__main__":
    main()

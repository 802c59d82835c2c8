#!/usr/bin/env python3
"""Golden vectors for ntlink_amd.anchor: the reference's own get_accepted_anchor_contigs
(bin/ntlink_utils.py:200-268), imported in the build container, on reads of the synthetic scenarios.
Output: tests/golden/gen/anchor_cases.json (data only).  Same import recipe as tests/golden/gen_goldens.py."""
import argparse
import gzip
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.modules["igraph"] = types.ModuleType("igraph")
sys.path.insert(0, "/root/reference/bin")
import ntlink_pair  # noqa: E402  (the reference)
import ntlink_utils  # noqa: E402

GEN = os.path.join(REPO, "tests", "golden", "gen")


def main():
    cases = []
    for name in ("syn_default", "syn_sensitive", "syn_x03", "syn_sens_repeat", "syn_many_ctg"):
        d = os.path.join(GEN, "synthetic")
        meta = json.load(open(os.path.join(d, name + ".json")))
        p = meta["params"]
        args = argparse.Namespace(k=meta["k"], z=p.get("z", 1000), x=p.get("x", 0.0), sensitive=p.get("sensitive", False))
        scaffolds = {n: ntlink_utils.Scaffold(id=n, length=l) for n, l in zip(meta["ctg_names"], meta["ctg_len"])}
        mx_info, dup = {}, set()
        for line in gzip.open(os.path.join(d, name + ".contigs.tsv.gz"), "rt"):
            f = line.strip().split("\t")
            if len(f) > 1:
                for tok in f[1].split(" "):
                    mx, pos, strand = tok.split(":")
                    if mx in mx_info:
                        dup.add(mx)
                    else:
                        mx_info[mx] = ntlink_pair.Minimizer(f[0], int(pos), strand)
        mx_info = {m: v for m, v in mx_info.items() if m not in dup}
        n_done = 0
        for ridx, line in enumerate(gzip.open(os.path.join(d, name + ".reads.tsv.gz"), "rt")):
            f = line.strip().split("\t")
            if len(f) < 3:
                continue
            mx_list = [(mx, int(pos), strand) for mx, pos, strand in (t.split(":") for t in f[2].split(" ")) if mx in mx_info]
            if not mx_list:
                continue
            acc, order = ntlink_utils.get_accepted_anchor_contigs(mx_list, int(f[1]), scaffolds, mx_info, args)
            cases.append({"scenario": name, "read_index": ridx, "read": f[0],
                          "order": order,
                          "hits": {c: [[h.mx, h.ctg_pos, h.ctg_strand, h.read_pos, h.read_strand] for h in acc[c].hits] for c in order}})
            n_done += 1
            if n_done >= 25:
                break
    json.dump(cases, open(os.path.join(GEN, "anchor_cases.json"), "w"), indent=0)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()

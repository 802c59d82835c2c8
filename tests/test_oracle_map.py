"""The oracle's index + mapping + pair tally against (a) the goldens the reference ships,
(b) vectors produced by importing the reference's Python (tests/golden/gen_goldens.py)."""
import os

import numpy as np
import pytest

import oracle
from helpers import (FIXTURES, GEN, REF, SCENARIOS, TEST7_PAF, contig_ids, load_scenario, parse_indexlr,
                     read_text)


def run_oracle(ctext, rtext, ctg_names, ctg_len, k, z=1000, a=1, f=10, x=0.0, sensitive=False,
               repeat_filter=False, threads=1):
    cn, _, coff, ch, cp, cs = parse_indexlr(ctext, False)
    name_to_id = {n: i for i, n in enumerate(ctg_names)}
    ids = np.array([name_to_id[n] for n in cn], np.uint32)
    cid = ids[contig_ids(coff)] if len(ch) else np.empty(0, np.uint32)
    index = oracle.Index(ch, cid, cp, cs)
    rn, rlen, roff, rh, rp, rs = parse_indexlr(rtext, True)
    res = oracle.map_reads(index, ctg_len, roff, rlen, rh, rp, rs, k=k, z=z, x=x, sensitive=sensitive,
                           repeat_filter=repeat_filter, threads=threads)
    verbose = oracle.format_verbose(res, rn, ctg_names)
    paf = oracle.format_paf(res, rn, rlen, ctg_names, ctg_len)
    by_name = dict(zip(ctg_names, (int(v) for v in ctg_len)))
    pairs = oracle.filter_pairs(oracle.tally_pairs(res, rlen, ctg_names, ctg_len, k, f), by_name, a)
    return verbose, paf, oracle.format_pairs(pairs), pairs, by_name


def sketch_text(path, k, w, with_len):
    recs = []
    for name, seq in oracle.read_fastx(path):
        h, p, s = oracle.sketch_seq(seq, k, w)
        recs.append((name, len(seq), h, p, s))
    return oracle.format_indexlr(recs, with_len), [r[0] for r in recs], [r[1] for r in recs]


@pytest.mark.parametrize("flags", [{}, {"sensitive": True}, {"repeat_filter": True}],
                         ids=["default", "sensitive", "repeat"])
@pytest.mark.parametrize("tag,target,reads,k,w,gold", FIXTURES)
def test_fixture_outputs(tag, target, reads, k, w, gold, flags):
    ctext, names, lens = sketch_text(os.path.join(REF, target), k, w, False)
    rtext, _, _ = sketch_text(os.path.join(REF, reads), k, w, True)
    verbose, paf, pairs_txt, pairs, by_name = run_oracle(ctext, rtext, names, np.array(lens, np.uint32), k,
                                                         threads=3, **flags)
    full = tag + "".join("." + f for f in flags)
    d = os.path.join(GEN, "fixtures")
    assert verbose == read_text(os.path.join(d, full + ".verbose_mapping.tsv"))
    assert paf == read_text(os.path.join(d, full + ".paf"))
    assert pairs_txt == read_text(os.path.join(d, full + ".pairs.tsv"))
    if gold and not flags:
        # goldens shipped by the reference itself
        exp = os.path.join(REF, "expected_outputs", gold + ".z1000")
        assert pairs_txt == read_text(exp + ".pairs.tsv")
        shipped = read_text(exp + ".verbose_mapping.tsv").splitlines()
        ours = set(verbose.splitlines())
        assert all(line in ours for line in shipped)  # tests 1-3 goldens are stale subsets
        if tag == "t4_k40_w100":
            assert verbose == read_text(exp + ".verbose_mapping.tsv")
        head, nodes, edges = oracle.format_dot(pairs, by_name, n=1)
        dot = read_text(exp + ".n1.scaffold.dot").splitlines(keepends=True)
        assert dot[:2] == head and dot[-1] == "}\n"
        body = dot[2:-1]
        assert set(l for l in body if "->" not in l) == nodes       # node order is Python-set order
        assert [l for l in body if "->" in l] == edges
    if tag.startswith("t7"):
        assert set(paf.splitlines()) == TEST7_PAF


@pytest.mark.parametrize("name", SCENARIOS)
def test_synthetic_scenarios(name):
    meta, ctext, rtext, exp = load_scenario(name)
    p = dict(meta["params"])
    verbose, paf, pairs_txt, _, _ = run_oracle(ctext, rtext, meta["ctg_names"],
                                               np.array(meta["ctg_len"], np.uint32), meta["k"],
                                               z=p.get("z", 1000), a=p.get("a", 1), f=p.get("f", 10),
                                               x=p.get("x", 0.0), sensitive=p.get("sensitive", False),
                                               repeat_filter=p.get("repeat_filter", False), threads=2)
    assert verbose == exp[".verbose_mapping.tsv"]
    assert paf == exp[".paf"]
    assert pairs_txt == exp[".pairs.tsv"]


def test_index_duplicates_removed():
    h = np.array([5, 7, 5, 9, 7, 11], np.uint64)
    ix = oracle.Index(h, np.arange(6, dtype=np.uint32), np.arange(6, dtype=np.uint32) * 10,
                      np.array([1, 0, 1, 0, 1, 0], np.uint8))
    assert len(ix) == 2
    assert ix.lookup(5) is None and ix.lookup(7) is None and ix.lookup(1) is None
    assert ix.lookup(9) == (3, 30, 0) and ix.lookup(11) == (5, 50, 0)

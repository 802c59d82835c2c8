"""The oracle's sketch (ntHash + Indexlr restatement) against the reference's golden indexlr output."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from helpers import FIXTURES, GEN, REF


@pytest.mark.parametrize("tag,target,reads,k,w,gold", [f for f in FIXTURES if f[5]])
def test_contig_tsv_matches_reference_golden(tag, target, reads, k, w, gold):
    recs = []
    for name, seq in oracle.read_fastx(os.path.join(REF, target)):
        h, p, s = oracle.sketch_seq(seq, k, w)
        recs.append((name, len(seq), h, p, s))
    text = oracle.format_indexlr(recs)
    assert text == open(os.path.join(REF, "expected_outputs", gold + ".tsv")).read()


@pytest.mark.parametrize("tag,target,reads,k,w,gold", FIXTURES)
def test_read_sketch_md5(tag, target, reads, k, w, gold):
    """Read sketches have no golden file of their own; their md5 is recorded when the mapping goldens
    are generated (tests/golden/gen_goldens.py) and agrees with SURVEY.md appendix A."""
    summ = json.load(open(os.path.join(GEN, "fixtures", "summary.json")))[tag]
    recs = []
    for name, seq in oracle.read_fastx(os.path.join(REF, reads)):
        h, p, s = oracle.sketch_seq(seq, k, w)
        recs.append((name, len(seq), h, p, s))
    text = oracle.format_indexlr(recs, with_len=True)
    assert hashlib.md5(text.encode()).hexdigest() == summ["read_tsv_md5"]


def test_batch_equals_per_sequence():
    seqs = [s for _, s in oracle.read_fastx(os.path.join(REF, "long_reads_4_top5.fa"))]
    off = np.zeros(len(seqs) + 1, np.uint64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    mx_off, h, p, s = oracle.sketch_batch(b"".join(seqs), off, 40, 100, threads=2)
    for i, q in enumerate(seqs):
        hh, pp, ss = oracle.sketch_seq(q, 40, 100)
        a, b = int(mx_off[i]), int(mx_off[i + 1])
        assert np.array_equal(h[a:b], hh) and np.array_equal(p[a:b], pp) and np.array_equal(s[a:b], ss)


def test_edge_cases():
    k, w = 8, 4
    assert len(oracle.sketch_seq(b"", k, w)[0]) == 0
    assert len(oracle.sketch_seq(b"ACGTACG", k, w)[0]) == 0            # shorter than k
    assert len(oracle.sketch_seq(b"ACGTACGTAC", k, w)[0]) == 0         # fewer than w k-mers
    assert len(oracle.sketch_seq(b"ACGTACGTACG", k, w)[0]) >= 1        # exactly w k-mers: one window
    assert len(oracle.sketch_seq(b"N" * 100, k, w)[0]) == 0
    # lowercase == uppercase; any non-ACGT byte is a break
    a = oracle.sketch_seq(b"ACGTTGCATGCATGCCGTAGCTAGCTAGGATC", k, w)
    b = oracle.sketch_seq(b"acgttgcatgcatgccgtagctagctaggatc", k, w)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # window counts valid k-mers, so an N run does not consume window slots
    left, right = b"ACGTTGCATGCATGCCGTAG", b"CTAGCTAGGATCCGATTACG"
    h, p, s = oracle.sketch_seq(left + b"NNNNN" + right, k, w)
    assert len(h) > 0 and all((q + k <= len(left)) or (q >= len(left) + 5) for q in p)
    # poly-A: every window's rightmost minimum is its last k-mer -> density 1
    h, p, s = oracle.sketch_seq(b"A" * 50, k, w)
    assert list(p) == list(range(w - 1, 50 - k + 1))

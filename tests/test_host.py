"""Host side of the product (no GPU): sequence input, text formats, pair tally / graph writers, CLI
argument surfaces.  Records come from the oracle here; the same writers are fed by the GPU in
tests/test_gpu_cli.py."""
import io
import os

import numpy as np
import pytest

import oracle
from helpers import FIXTURES, GEN, REF, SCENARIOS, contig_ids, load_scenario, parse_indexlr, read_text
from ntlink_amd import cli, formats, pairing, seqio


@pytest.mark.parametrize("fname", ["scaffolds_2.fa", "long_reads_2.fq.gz", "long_reads_3.fa.gz", "long_reads_4_top5.fa"])
def test_seqio_matches_reference_reader_semantics(fname):
    a = list(seqio.read_fastx(os.path.join(REF, fname)))
    b = list(oracle.read_fastx(os.path.join(REF, fname)))
    assert a == b and len(a) > 0
    ss = seqio.load_all([os.path.join(REF, fname)])
    assert ss.names == [n for n, _ in b] and ss.bases == sum(len(s) for _, s in b)


def test_seqio_edge_records(tmp_path):
    p = tmp_path / "x.fa"
    p.write_text(">a desc here\nACGT\nAC\n>b\n\n>c\tz\nGG\n@q1 x\nACGT\n+\nIIII\n@q2\nAC\nGT\n+q2\nII\nII\n>d\nTT")
    recs = list(seqio.read_fastx(str(p)))
    assert recs == [("a", b"ACGTAC"), ("b", b""), ("c", b"GG"), ("q1", b"ACGT"), ("q2", b"ACGT"), ("d", b"TT")]
    for env in ({}, {"NTL_IO_NO_MMAP": "1"}):  # parallel pread source and the serial (pipe / zlib) source
        with _env(env):
            batches = list(seqio.load([str(p)], max_bases=6))
        assert all(len(b) > 0 for b in batches) and len(batches) >= 2
        assert _records(batches) == recs


class _env:
    def __init__(self, kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _records(batches):
    out = []
    for b in batches:
        raw, off = b.buf.tobytes(), b.offsets.tolist()
        out += [(n, raw[off[i]:off[i + 1]]) for i, n in enumerate(b.names)]
    return out


def _random_fastx(rng, nrec, fastq, multiline, crlf=False):
    """Text with the layouts the record state machine has to survive: wrapped sequences, empty
    records, quality lines that start with '@' or '>', headers with descriptions."""
    nl = "\r\n" if crlf else "\n"
    out = []
    for i in range(nrec):
        n = int(rng.integers(0, 400)) if rng.random() < 0.9 else int(rng.integers(2000, 9000))
        seq = "".join(rng.choice(list("ACGTNacgt"), n))
        wrap = int(rng.integers(20, 90)) if multiline else max(n, 1)
        lines = [seq[j:j + wrap] for j in range(0, n, wrap)] or [""]
        hdr = f"r{i}" + (" some description" if rng.random() < 0.3 else "")
        if fastq:
            q = "".join(rng.choice(list("@>+!IJ5"), n))
            qlines = [q[j:j + wrap] for j in range(0, n, wrap)] or [""]
            out += ["@" + hdr] + lines + ["+" + (hdr if rng.random() < 0.2 else "")] + qlines
        else:
            out += [">" + hdr] + lines
    return nl.join(out) + (nl if rng.random() < 0.7 else "")


@pytest.mark.parametrize("fastq,multiline", [(False, True), (False, False), (True, False), (True, True)])
def test_parallel_reader_equals_serial_semantics(tmp_path, fastq, multiline):
    """Byte ranges cut at guessed record boundaries and parsed by many threads give the records of
    the one-pass reader (bin/read_fasta.py:6-46), also when a guess lands inside a quality section."""
    rng = np.random.default_rng(5 + 2 * fastq + multiline)
    for rep in range(3):
        p = tmp_path / f"x{rep}.{'fq' if fastq else 'fa'}"
        p.write_text(_random_fastx(rng, 6000, fastq, multiline, crlf=(rep == 2)), newline="")
        want = list(seqio.read_fastx(str(p)))
        assert len(want) == 6000
        with _env({"NTL_IO_THREADS": "13", "NTL_IO_MIN_CHUNK": "3000"}):
            whole = seqio.load_all([str(p)])
            parts = list(seqio.load([str(p)], max_bases=300_000))
        assert _records([whole]) == want
        assert len(parts) > 2 and _records(parts) == want
        with _env({"NTL_IO_NO_MMAP": "1"}):
            assert _records(list(seqio.load([str(p)], max_bases=300_000))) == want


@pytest.mark.parametrize("tag,target,reads,k,w,gold", [f for f in FIXTURES if f[5]])
def test_indexlr_writer_and_parser(tag, target, reads, k, w, gold):
    ss = seqio.load_all([os.path.join(REF, target)])
    off, h, p, s = oracle.sketch_batch(ss.buf, ss.offsets, k, w)
    buf = io.StringIO()
    formats.write_indexlr(buf, ss.names, ss.lengths, off, h, p, s, False)
    text = buf.getvalue()
    assert text == read_text(os.path.join(REF, "expected_outputs", gold + ".tsv"))
    n2, _, o2, h2, p2, s2 = formats.parse_indexlr(io.StringIO(text), False)
    assert n2 == ss.names and np.array_equal(o2, off) and np.array_equal(h2, h) and np.array_equal(p2, p) and np.array_equal(s2, s)


def _oracle_records(target, reads, k, w, **kw):
    cs_, rs_ = seqio.load_all([os.path.join(REF, target)]), seqio.load_all([os.path.join(REF, reads)])
    coff, ch, cp, cst = oracle.sketch_batch(cs_.buf, cs_.offsets, k, w)
    roff, rh, rp, rst = oracle.sketch_batch(rs_.buf, rs_.offsets, k, w)
    ix = oracle.Index(ch, contig_ids(coff), cp, cst)
    res = oracle.map_reads(ix, cs_.lengths, roff, rs_.lengths, rh, rp, rst, k=k, z=1000, threads=0, **kw)
    return cs_, rs_, res


@pytest.mark.parametrize("tag,target,reads,k,w,gold", FIXTURES)
def test_writers_and_tally_on_fixtures(tag, target, reads, k, w, gold, tmp_path):
    cs_, rs_, res = _oracle_records(target, reads, k, w)
    d = os.path.join(GEN, "fixtures")
    v, p = io.StringIO(), io.StringIO()
    formats.write_verbose(v, res, rs_.names, cs_.names)
    formats.write_paf(p, res, rs_.names, rs_.lengths, cs_.names, cs_.lengths)
    assert v.getvalue() == read_text(os.path.join(d, tag + ".verbose_mapping.tsv"))
    assert p.getvalue() == read_text(os.path.join(d, tag + ".paf"))
    t = pairing.PairTally(cs_.names, cs_.lengths, k, 10)
    # feed in two batches: the tally is order-sensitive and must not care about batching
    half = len(rs_) // 2
    cut = int(np.searchsorted(res["maps"]["read"], half))
    hcut = int(res["maps"]["hit_off"][cut]) if cut < len(res["maps"]) else len(res["hits"])
    m1 = res["maps"][:cut]
    m2 = res["maps"][cut:].copy()
    m2["hit_off"] -= hcut
    t.add_batch({"maps": m1, "hits": res["hits"][:hcut]}, rs_.lengths)
    t.add_batch({"maps": m2, "hits": res["hits"][hcut:]}, rs_.lengths)
    pairs = t.filtered(1)
    buf = io.StringIO()
    pairing.write_pairs(buf, pairs)
    assert buf.getvalue() == read_text(os.path.join(d, tag + ".pairs.tsv"))
    if gold:
        dot = io.StringIO()
        pairing.write_dot(dot, pairs, cs_.names, cs_.lengths, 1)
        got = dot.getvalue().splitlines()
        exp = read_text(os.path.join(REF, "expected_outputs", gold + ".z1000.n1.scaffold.dot")).splitlines()
        assert got[:2] == exp[:2] and got[-1] == exp[-1] == "}"
        assert sorted(l for l in got if "->" not in l) == sorted(l for l in exp if "->" not in l)
        assert [l for l in got if "->" in l] == [l for l in exp if "->" in l]
    # the native writers (ntl_tally_write, what the drivers use) leave the bytes of the Python forms, for every filter setting
    for a, min_n in ((1, 1), (2, 1), (1, 2), (3, 3)):
        kept = t.write(a, min_n, str(tmp_path / "n.pairs.tsv"), str(tmp_path / "n.dot"))
        fp = t.filtered(a)
        assert kept == len(fp)
        pb, db = io.StringIO(), io.StringIO()
        pairing.write_pairs(pb, fp)
        pairing.write_dot(db, fp, cs_.names, cs_.lengths, min_n)
        assert read_text(str(tmp_path / "n.pairs.tsv")) == pb.getvalue(), (a, min_n)
        assert read_text(str(tmp_path / "n.dot")) == db.getvalue(), (a, min_n)


@pytest.mark.parametrize("name", SCENARIOS)
def test_tally_on_synthetic_scenarios(name):
    meta, ctext, rtext, exp = load_scenario(name)
    p = meta["params"]
    cn, _, coff, ch, cp, cs = parse_indexlr(ctext, False)
    ids = np.array([meta["ctg_names"].index(n) for n in cn], np.uint32)
    ix = oracle.Index(ch, ids[contig_ids(coff)], cp, cs)
    rn, rlen, roff, rh, rp, rs = parse_indexlr(rtext, True)
    cl = np.array(meta["ctg_len"], np.uint32)
    res = oracle.map_reads(ix, cl, roff, rlen, rh, rp, rs, k=meta["k"], z=p.get("z", 1000), x=p.get("x", 0.0),
                           sensitive=p.get("sensitive", False), repeat_filter=p.get("repeat_filter", False))
    t = pairing.PairTally(meta["ctg_names"], cl, meta["k"], p.get("f", 10))
    t.add_batch(res, rlen)
    buf = io.StringIO()
    pairing.write_pairs(buf, t.filtered(p.get("a", 1)))
    assert buf.getvalue() == exp[".pairs.tsv"]


@pytest.mark.parametrize("tag,target,k", [("t4_k40_w100", "scaffolds_4.fa", 40), ("t3_k24_w250", "scaffolds_3.fa", 24)])
def test_checkpoint_retally(tag, target, k):
    """A pre-existing verbose_mapping.tsv bypasses mapping (bin/ntlink_pair.py:565-575, 437-488)."""
    cs_ = seqio.load_all([os.path.join(REF, target)])
    t = pairing.PairTally(cs_.names, cs_.lengths, k, 10)
    index_of = {n: i for i, n in enumerate(cs_.names)}
    with open(os.path.join(GEN, "fixtures", tag + ".verbose_mapping.tsv")) as fh:
        for _read, entries in formats.parse_verbose(fh):
            t.add_checkpoint_read(entries, index_of)
    buf = io.StringIO()
    pairing.write_pairs(buf, t.filtered(1))
    assert buf.getvalue() == read_text(os.path.join(GEN, "fixtures", tag + ".checkpoint.pairs.tsv"))


def test_cli_surfaces():
    a = cli.ntlink_pair_parser().parse_args("-p out -n 1 -m c.tsv -s t.fa -k 32 -a 1 -z 1000 -f 10 -x 0 --verbose --paf -".split())
    assert (a.FILES, a.z, a.x, a.paf, a.verbose, a.pairs, a.repeat_filter) == (["-"], 1000, 0.0, True, True, False, False)
    assert cli.ntlink_pair_parser().parse_args("-m c -s t -k 32 r.tsv".split()).z == 500  # argparse default differs from make's
    assert cli.ntlink_main(["help"]) == 0
    assert cli.ntlink_main(["scaffold", "target=a", "reads=b"]) == 2
    assert cli.ntlink_main(["pair", "-B", "target=a"]) == 2


@pytest.mark.parametrize("tag,target,reads,k,w,gold", [FIXTURES[1], FIXTURES[3]])
def test_native_writers_equal_python_writers(tag, target, reads, k, w, gold, tmp_path):
    """csrc/ntl_io.cpp emitters (real file descriptors) == the Python emitters == the goldens."""
    cs_, rs_, res = _oracle_records(target, reads, k, w)
    d = os.path.join(GEN, "fixtures")
    with open(tmp_path / "v.tsv", "w") as fh:
        formats.write_verbose(fh, res, rs_.names, cs_.names)
    with open(tmp_path / "p.paf", "w") as fh:
        formats.write_paf(fh, res, rs_.names, rs_.lengths, cs_.names, cs_.lengths)
    assert read_text(str(tmp_path / "v.tsv")) == read_text(os.path.join(d, tag + ".verbose_mapping.tsv"))
    assert read_text(str(tmp_path / "p.paf")) == read_text(os.path.join(d, tag + ".paf"))
    off, h, p, s = oracle.sketch_batch(cs_.buf, cs_.offsets, k, w)
    with open(tmp_path / "c.tsv", "w") as fh:
        formats.write_indexlr(fh, cs_.names, cs_.lengths, off, h, p, s, False)
    assert read_text(str(tmp_path / "c.tsv")) == read_text(os.path.join(REF, "expected_outputs", gold + ".tsv"))
    roff, rh, rp, rst = oracle.sketch_batch(rs_.buf, rs_.offsets, k, w)
    with open(tmp_path / "r.tsv", "w") as fh:
        formats.write_indexlr(fh, rs_.names, rs_.lengths, roff, rh, rp, rst, True)
    buf = io.StringIO()
    formats.write_indexlr(buf, rs_.names.tolist(), rs_.lengths, roff, rh, rp, rst, True)
    assert read_text(str(tmp_path / "r.tsv")) == buf.getvalue()


def test_native_reader_batches_and_stdin(tmp_path):
    import subprocess
    import sys
    src = os.path.join(REF, "long_reads_2.fq.gz")
    whole = seqio.load_all([src])
    parts = list(seqio.load([src], max_bases=500_000))
    assert len(parts) > 3 and sum(len(p) for p in parts) == len(whole)
    assert b"".join(p.buf.tobytes() for p in parts) == whole.buf.tobytes()
    assert [n for p in parts for n in p.names] == whole.names.tolist()
    code = ("import sys; sys.path.insert(0, %r); from ntlink_amd import seqio; s = seqio.load_all(['-']); "
            "print(len(s), s.bases)" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    with open(src, "rb") as fin:
        out = subprocess.check_output([sys.executable, "-c", code], stdin=fin).decode().split()
    assert [int(v) for v in out] == [len(whole), whole.bases]


def test_native_tsv_parser_equals_python_parser(tmp_path):
    """ntl_tsv_* (blocks of lines on several threads) == formats.parse_indexlr == what the reference's
    split()s see (bin/ntlink_pair.py:197-207,355-378): blank lines, records without minimizers,
    CRLF, a last line without newline, blocks smaller than a line."""
    rng = np.random.default_rng(3)
    lines = []
    for i in range(3000):
        n = int(rng.integers(0, 40)) if rng.random() < 0.9 else 0
        toks = " ".join(f"{int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2))}:{int(rng.integers(0, 2**31))}:{'+-'[int(rng.integers(0, 2))]}"
                        for _ in range(n))
        lines.append(f"read_{i}\t{int(rng.integers(1, 10**6))}\t{toks}")
        if rng.random() < 0.02:
            lines.append("")
    for with_len, nl, tail in ((True, "\n", "\n"), (True, "\r\n", ""), (False, "\n", "")):
        body = [l if with_len else l.split("\t")[0] + "\t" + l.split("\t")[2] if l else l for l in lines]
        p = tmp_path / f"x_{with_len}_{len(nl)}.tsv"
        p.write_text(nl.join(body) + tail, newline="")
        with open(p) as fh:
            want = formats.parse_indexlr(fh, with_len)
        for max_bytes, thr in ((0, "7"), (5000, "3"), (100, "1")):
            with _env({"NTL_IO_THREADS": thr, "NTL_IO_MIN_CHUNK": "700"}):
                parts = list(formats.read_indexlr(str(p), with_len, max_bytes))
            assert [n for x in parts for n in x[0]] == want[0]
            if with_len:
                assert np.array_equal(np.concatenate([x[1] for x in parts]), want[1])
            assert np.array_equal(np.concatenate([np.diff(x[2]) for x in parts]), np.diff(want[2]))
            for col in (3, 4, 5):
                assert np.array_equal(np.concatenate([x[col] for x in parts]), want[col])
            if max_bytes:
                assert len(parts) > 3
    bad = tmp_path / "bad.tsv"
    bad.write_text("r1\t100\t12:5:+ 13:x:-\n")
    with pytest.raises(ValueError):
        list(formats.read_indexlr(str(bad), True))
    # `indexlr --pos` without `--strand` (the overlap stage's TSV, ntLink:244,249): H:pos tokens
    pos = tmp_path / "pos.tsv"
    pos.write_text("c1\t18446744073709551615:0 7:12 99:4000000000\nc2\t\nc3\t5:6\n")
    (names, _l, off, h, p, s), = list(formats.read_indexlr(str(pos), False, with_strand=False))
    assert list(names) == ["c1", "c2", "c3"] and off.tolist() == [0, 3, 3, 4]
    assert h.tolist() == [18446744073709551615, 7, 99, 5] and p.tolist() == [0, 12, 4000000000, 6] and s.tolist() == [1, 1, 1, 1]
    for text in ("c1\t5:6:+\n", "c1\t5\n", "c1\t5:\n"):
        bad.write_text(text)
        with pytest.raises(ValueError):
            list(formats.read_indexlr(str(bad), False, with_strand=False))


def test_prefetch_and_drain_keep_order_and_propagate_errors():
    from ntlink_amd.pipeline import Drain, Prefetch
    assert list(Prefetch(iter(range(50)), depth=2)) == list(range(50))

    def boom():
        yield 1
        raise OSError("reader failed")
    it = iter(Prefetch(boom()))
    assert next(it) == 1
    with pytest.raises(OSError):
        next(it)
    got = []
    d = Drain(lambda a, b: got.append((a, b)))
    for i in range(20):
        d.put(i, -i)
    d.close()
    assert got == [(i, -i) for i in range(20)]

    def bad(x):
        if x == 3:
            raise ValueError("writer failed")
    d = Drain(bad)
    with pytest.raises(ValueError):
        for i in range(10):
            d.put(i)
        d.close()
    # an abandoned stream stops its producer
    closed = []

    def gen():
        try:
            for i in range(10**6):
                yield i
        finally:
            closed.append(True)
    p = Prefetch(gen(), depth=1)
    assert next(iter(p)) == 0
    p.stop()
    assert closed == [True]


def test_gzip_sources_members_and_truncation(tmp_path):
    """`gzip -cd -f` semantics of ntLink:113-117: concatenated members are one stream, whether the file is
    inflated in one go (libdeflate / zlib) or streamed through inflate(); a cut-off file is an error."""
    import gzip
    rng = np.random.default_rng(21)
    text = _random_fastx(rng, 800, True, False).encode()
    cut = text.index(b"\n@", len(text) // 2) + 1
    p = tmp_path / "two_members.fq.gz"
    p.write_bytes(gzip.compress(text[:cut]) + gzip.compress(text[cut:]))
    plain = tmp_path / "plain.fq"
    plain.write_bytes(text)
    want = list(seqio.read_fastx(str(plain)))
    assert len(want) == 800
    for env in ({}, {"NTL_IO_GZ_WHOLE_MAX": "0"}, {"NTL_IO_NO_LIBDEFLATE": "1"}, {"NTL_IO_NO_MMAP": "1"}):
        with _env(dict(env, NTL_IO_THREADS="4", NTL_IO_MIN_CHUNK="5000")):
            assert _records(list(seqio.load([str(p)], max_bases=40_000))) == want, env
    bad = tmp_path / "cut.fq.gz"
    bad.write_bytes(gzip.compress(text)[:-2000])
    for env in ({}, {"NTL_IO_GZ_WHOLE_MAX": "0"}):
        with _env(env), pytest.raises(OSError):
            list(seqio.load([str(bad)], max_bases=40_000))


def _bgzf(data, block=60000):
    """bgzip's container: gzip members with a 'BC' extra field (compressed size - 1), then the empty EOF member."""
    import struct
    import zlib
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        chunk = b"" if i is None else data[i:i + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = c.compress(chunk) + c.flush()
        bsize = 18 + len(raw) + 8
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize - 1) + raw +
                   struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


def test_bgzf_members_inflate_in_parallel(tmp_path):
    """bgzip-compressed reads: the member table comes from the 'BC' fields, members inflate on several threads;
    a file that only starts as BGZF falls back to the general path."""
    import gzip
    rng = np.random.default_rng(33)
    text = _random_fastx(rng, 3000, True, False).encode()
    plain = tmp_path / "plain.fq"
    plain.write_bytes(text)
    want = list(seqio.read_fastx(str(plain)))
    p = tmp_path / "reads.fq.gz"
    p.write_bytes(_bgzf(text))
    assert gzip.decompress(p.read_bytes()) == text  # the container is valid gzip
    with _env({"NTL_IO_THREADS": "6", "NTL_IO_MIN_CHUNK": "5000"}):
        assert _records(list(seqio.load([str(p)], max_bases=200_000))) == want
    mixed = tmp_path / "mixed.fq.gz"
    cut = text.index(b"\n@", len(text) // 2) + 1
    mixed.write_bytes(_bgzf(text[:cut])[:-28] + gzip.compress(text[cut:]))  # BGZF members, then an ordinary one
    assert _records(list(seqio.load([str(mixed)], max_bases=200_000))) == want


def test_duplicate_contig_ids_take_the_last_length(tmp_path):
    """The reference keeps one Scaffold per id, the last record's (dict overwrite, bin/ntlink_utils.py:65-73)."""
    from ntlink_amd.pipeline import _contig_lengths
    fa = tmp_path / "dup.fa"
    fa.write_text(">a\nACGTACGT\n>b\nAC\n>a\nACG\n>c\nA\n>b\nACGTA\n")
    names, ctg_len, index_of = _contig_lengths(str(fa))
    assert names == ["a", "b", "a", "c", "b"] and ctg_len.tolist() == [3, 5, 3, 1, 5]
    assert index_of == {"a": 2, "b": 4, "c": 3}


def test_tally_overhang_check_only_for_evaluated_pairs_and_merge():
    """calculate_gap_size asserts a >= 0 and b >= 0 for the pairs it evaluates (bin/ntlink_pair.py:173-184): a read with
    one mapping, however odd, raises nothing.  merge(): two tallies over consecutive read ranges == one tally over both."""
    from ntlink_amd import capi
    from ntlink_amd.pairing import PairTally
    names, lens, k = ["c1", "c2", "c3"], [1000, 2000, 50], 32

    def recs(reads):
        maps, hits = [], []
        for r, ms in enumerate(reads):
            for ctg, hs in ms:
                maps.append((r, ctg, len(hs), 0, len(hits)))
                hits += [(cp, rp, cs, rs, (0, 0)) for cp, rp, cs, rs in hs]
        return {"maps": np.array(maps, capi.MAPPING_DT), "hits": np.array(hits, capi.HIT_DT)}
    # contig c3 is 50 long: a hit at 40 with k = 32 has a negative overhang behind it
    lone = [[(2, [(40, 100, 1, 1)])]]
    t = PairTally(names, lens, k)
    t.add_batch(recs(lone), [5000])
    assert t.pairs == {}
    with pytest.raises(AssertionError):
        t.add_batch(recs([[(2, [(40, 100, 1, 1)]), (0, [(10, 900, 1, 1)])]]), [5000])
    reads = [[(0, [(900, 100, 1, 1), (950, 150, 1, 1)]), (1, [(10, 400, 1, 1), (60, 450, 1, 1)])],
             [(1, [(1900, 100, 1, 1)]), (2, [(5, 300, 1, 1)])],
             [(0, [(800, 50, 1, 1), (850, 100, 1, 1)]), (1, [(5, 300, 1, 1), (55, 350, 1, 1)])]]
    rl = [6000, 7000, 8000]
    whole = PairTally(names, lens, k)
    whole.add_batch(recs(reads), rl)
    a, b = PairTally(names, lens, k), PairTally(names, lens, k)
    a.add_batch(recs(reads[:1]), rl[:1])
    b.add_batch(recs(reads[1:]), rl[1:])
    a.merge(b.export())
    assert list(a.pairs.items()) == list(whole.pairs.items()) and len(whole.pairs) >= 2


def test_plain_files_parse_from_a_mapping_or_staged_preads(tmp_path, monkeypatch):
    """A plain file is parsed from a mapping of the page cache, or (NTL_IO_PREAD=1) from staged preads; same records."""
    rng = np.random.default_rng(11)
    p = tmp_path / "asm.fa"
    with open(p, "w") as fh:
        for i in range(300):
            s = "".join(rng.choice(list("ACGTNacgt"), int(rng.integers(1, 5000))))
            fh.write(f">c{i} x\n" + "\n".join(s[j:j + 70] for j in range(0, len(s), 70)) + "\n")
    monkeypatch.setenv("NTL_IO_PREAD", "1")
    a = seqio.load_all([str(p)])
    a2 = seqio.concat(list(seqio.load([str(p)], max_bases=40_000)))
    monkeypatch.delenv("NTL_IO_PREAD")
    monkeypatch.setenv("NTL_IO_THREADS", "5")
    monkeypatch.setenv("NTL_IO_MIN_CHUNK", "3000")
    b = seqio.load_all([str(p)])
    b2 = seqio.concat(list(seqio.load([str(p)], max_bases=40_000)))
    for x in (a2, b, b2):
        assert a.names == x.names and np.array_equal(a.offsets, x.offsets) and np.array_equal(a.buf, x.buf) and len(a) == 300
    want = list(seqio.read_fastx(str(p)))
    assert [n for n, _ in want] == b.names.tolist() and b"".join(s for _, s in want) == b.buf.tobytes()


def _expected_pack(buf, off):
    """numpy restatement of pack_kernel / run_*_kernel (csrc/pack_kernels.h): 2-bit codes at global position 16 + b, ACGT runs"""
    total = int(off[-1])
    nwords = (16 + total + 4096 + 15) // 16 + 2
    code = np.zeros(256, np.uint64); ok = np.zeros(256, bool)
    for ch, v in zip(b"ACGTacgt", (0, 1, 2, 3, 0, 1, 2, 3)):
        code[ch] = v; ok[ch] = True
    vals = np.zeros(nwords * 16, np.uint64)
    vals[16:16 + total] = code[buf[:total]]
    words = (vals.reshape(-1, 16) << (2 * np.arange(16, dtype=np.uint64))).sum(axis=1).astype(np.uint32)
    valid = ok[buf[:total]]
    srf, rst, rln = [0], [], []
    for i in range(len(off) - 1):
        v = valid[int(off[i]):int(off[i + 1])]
        d = np.diff(np.concatenate(([0], v.astype(np.int8), [0])))
        st, en = np.flatnonzero(d == 1), np.flatnonzero(d == -1)
        rst += st.tolist(); rln += (en - st).tolist()
        srf.append(len(rst))
    return words, np.array(srf, np.uint32), np.array(rst, np.uint32), np.array(rln, np.uint32)


@pytest.mark.parametrize("threads,chunk,simd", [("1", "1000000", None), ("7", "900", None), ("1", "1000000", "avx2"), ("3", "5000", "none")])
def test_parser_packs_bases_and_runs_like_the_device(tmp_path, monkeypatch, threads, chunk, simd):
    """seqio.load(packed=True): 2-bit words and ACGT-run table made by the parser threads == what the device-side pack kernels
    derive from the ASCII bytes (range cuts at any 2-bit offset, N / IUPAC / lower case, wrapped lines, empty records); with every
    step width of the packer the CPU has (64 bases with AVX-512BW, 32 with AVX2 + BMI2, 8: NTL_IO_SIMD)."""
    if simd:
        monkeypatch.setenv("NTL_IO_SIMD", simd)
    rng = np.random.default_rng(int(threads))
    p = tmp_path / "x.fa"
    with open(p, "w") as fh:
        for i in range(400):
            n = int(rng.integers(0, 700)) if i % 9 else int(rng.integers(0, 4))
            s = "".join(rng.choice(list("ACGTACGTACGTacgtNnRYK-"), n)) if i % 3 else "".join(rng.choice(list("ACGT"), n))
            if i % 11 == 0:
                s = "N" * n
            fh.write(f">s{i}\n" + ("\n".join(s[j:j + 61] for j in range(0, n, 61)) if i % 2 else s) + "\n")
    monkeypatch.setenv("NTL_IO_THREADS", threads)
    monkeypatch.setenv("NTL_IO_MIN_CHUNK", chunk)
    plain = seqio.load_all([str(p)])
    for max_bases, one_pass in ((None, "0"), (5000, "0"), (5000, "1")):
        monkeypatch.setenv("NTL_IO_ONE_PASS", one_pass)
        sets = list(seqio.load([str(p)], max_bases=max_bases, packed=True))
        at = 0
        for ss in sets:
            n = len(ss)
            o0 = int(plain.offsets[at])
            sub_off = plain.offsets[at:at + n + 1] - np.uint64(o0)
            assert np.array_equal(ss.offsets, sub_off) and ss.names.tolist() == plain.names.tolist()[at:at + n] and ss.buf is None
            words, srf, rst, rln = _expected_pack(plain.buf[o0:int(plain.offsets[at + n])], sub_off)
            assert np.array_equal(ss.seq_run_first, srf) and np.array_equal(ss.run_start, rst) and np.array_equal(ss.run_len, rln)
            if one_pass == "0":
                assert ss.positions is None and np.array_equal(ss.packed, words)
            else:
                # one pass: the sequences of a parser thread sit at an upper bound of their place; every sequence's bases,
                # read back from its position in the stream, are those of the contiguous layout; positions are in order, word
                # starts of the threads' segments are multiples of 16, nothing overlaps and everything lies inside the stream
                assert ss.positions is not None and len(ss.packed) == (16 + ss.span_positions + 4096 + 15) // 16 + 2
                two = np.frombuffer(ss.packed.tobytes(), np.uint8)
                codes = np.stack([(two >> s) & 3 for s in (0, 2, 4, 6)], axis=1).reshape(-1)
                exp = np.frombuffer(words.tobytes(), np.uint8)
                exp_codes = np.stack([(exp >> s) & 3 for s in (0, 2, 4, 6)], axis=1).reshape(-1)
                ends = ss.positions + ss.lengths
                assert (ss.positions[1:] >= ends[:-1]).all() and (not n or int(ends[-1]) <= ss.span_positions)
                for i in range(n):
                    a, ln = 16 + int(ss.positions[i]), int(ss.lengths[i])
                    b = 16 + int(sub_off[i])
                    assert np.array_equal(codes[a:a + ln], exp_codes[b:b + ln]), (i, ln)
            at += n
        assert at == len(plain) == 400


def test_nthash_ring_sums_cannot_wrap():
    """sketch_fast_kernel orders k-mers by the sum of the 31-bit rings (bits 33..63) of fwd and rev and needs that sum never to
    be 2^31 - 1 (the carry of the low 33 bits would wrap it to 0).  F + R = 2^31 - 1 means F ^ R is all ones: 31 bits, odd.
    F ^ R is the XOR of rotated seed rings, one of a base and one of its complement per position, and those pairs have an even
    number of set bits between them; checked on the seeds and, through the oracle's hashes, on random k-mers."""
    seeds = {"A": 0x3c8bfbb395c60474, "C": 0x3193c18562a02b4c, "G": 0x20323ed082572324, "T": 0x295549f54be24456}
    ring = {b: s >> 33 for b, s in seeds.items()}
    assert (bin(ring["A"]).count("1") + bin(ring["T"]).count("1")) % 2 == 0
    assert (bin(ring["C"]).count("1") + bin(ring["G"]).count("1")) % 2 == 0

    def rot31(x, n):
        n %= 31
        return ((x << n) | (x >> (31 - n))) & 0x7FFFFFFF

    rng = np.random.default_rng(4)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for k in (7, 16, 24, 32, 33, 100):
        text = "".join(rng.choice(list("ACGT"), k + 300))
        h0 = oracle.hash_seq(text.encode(), k)[0]
        for p in range(0, 300, 7):
            f = r = 0
            for i, b in enumerate(text[p:p + k]):
                f ^= rot31(ring[b], k - 1 - i)
                r ^= rot31(ring[comp[b]], i)
            assert bin(f ^ r).count("1") % 2 == 0
            c = int(h0[p]) >> 33
            assert c in ((f + r) & 0x7FFFFFFF, (f + r + 1) & 0x7FFFFFFF) and (f + r) & 0x7FFFFFFF != 0x7FFFFFFF


def test_parallel_readers_keep_the_input_order(tmp_path):
    """seqio.load_parallel: several readers on chunks of the input (byte ranges of plain and BGZF files, whole gzip streams),
    batches handed out in input order -- the records and the byte accounting of one serial reader, for any reader count and
    chunk size; a consumer that stops early leaves no thread behind."""
    import gzip
    import threading
    from helpers import write_bgzf
    rng = np.random.default_rng(5)
    paths = []
    for f in range(3):
        p = tmp_path / f"r{f}.fa"
        with open(p, "w") as fh:
            for i in range(300):
                fh.write(f">f{f}_{i} c\n" + "".join(rng.choice(list("ACGTN"), int(rng.integers(1, 5000)))) + "\n")
        paths.append(str(p))
    write_bgzf(str(tmp_path / "z.fa.gz"), open(paths[1], "rb").read())
    with gzip.open(tmp_path / "plain.fa.gz", "wt") as fh:
        fh.write(">g1\nACGTACGT\n>g2\nTTTT\n")
    paths[1:1] = [str(tmp_path / "z.fa.gz"), str(tmp_path / "plain.fa.gz")]
    want = _records(list(seqio.load(paths, max_bases=30_000)))
    assert len(want) == 1202
    total = sum(os.path.getsize(p) for p in paths)
    before = threading.active_count()
    for readers in (2, 3, 5):
        for chunk in (50_000, 300_000, 10_000_000):
            st = {}
            got = _records(list(seqio.load_parallel(paths, readers=readers, chunk_bytes=chunk, max_bases=30_000, stats=st)))
            assert got == want, (readers, chunk)
            assert st["parsed_bytes"] == total
    g = seqio.load_parallel(paths, readers=3, chunk_bytes=50_000, max_bases=30_000)
    next(g)
    g.close()
    assert threading.active_count() <= before + 1
    # several gzip streams in a row are one chunk (inflated ahead by one reader's `load`), plain files between and behind them
    # are cut as before: same records, same order
    gz = []
    for f in range(5):
        q = tmp_path / f"s{f}.fq.gz"
        with gzip.open(q, "wt") as fh:
            for i in range(40):
                sq = "".join(rng.choice(list("ACGT"), int(rng.integers(1, 3000))))
                fh.write(f"@s{f}_{i}\n{sq}\n+\n{'I' * len(sq)}\n")
        gz.append(str(q))
    mixed = gz[:3] + [paths[0]] + gz[3:] + [paths[-1]]
    want = _records(list(seqio.load(mixed, max_bases=30_000)))
    for readers in (2, 3):
        assert _records(list(seqio.load_parallel(mixed, readers=readers, chunk_bytes=100_000, max_bases=30_000))) == want
    assert _records(list(seqio.load_parallel(gz, readers=2, max_bases=30_000))) == _records(list(seqio.load(gz, max_bases=30_000)))


def test_native_pair_writers_on_made_up_tallies(tmp_path):
    """ntl_tally_write against the Python writers on pairs the fixtures do not have: even and odd gap lists with negative medians
    (int() truncates toward zero), estimates at and beyond minus a contig's length (dropped), contigs called ntLink_N (scaf_num),
    a pair whose reverse complement is another pair's edge (the dict keeps the first place and the later value)."""
    rng = np.random.default_rng(9)
    names = [f"ntLink_{i}" if i % 3 == 0 else f"c{i}" for i in range(40)]
    lens = rng.integers(50, 400, len(names)).astype(np.uint32)
    t = pairing.PairTally(names, lens, 32, 10)
    n = 300
    src, tgt = rng.integers(0, len(names), n).astype(np.uint32), rng.integers(0, len(names), n).astype(np.uint32)
    keep = src != tgt
    src, tgt = src[keep], tgt[keep]
    so, to = rng.integers(0, 2, len(src)).astype(np.uint8), rng.integers(0, 2, len(src)).astype(np.uint8)
    anchor = rng.integers(1, 5, len(src)).astype(np.uint32)
    ng = rng.integers(1, 6, len(src))
    goff = np.zeros(len(src) + 1, np.uint64)
    goff[1:] = np.cumsum(ng)
    gaps = rng.integers(-500, 500, int(goff[-1])).astype(np.int64)
    t.merge((src, so, tgt, to, anchor, goff, gaps))
    for a, min_n in ((1, 1), (2, 2), (4, 1)):
        kept = t.write(a, min_n, str(tmp_path / "p.tsv"), str(tmp_path / "g.dot"))
        fp = t.filtered(a)
        assert 0 < kept == len(fp) < len(t.pairs)
        pb, db = io.StringIO(), io.StringIO()
        pairing.write_pairs(pb, fp)
        pairing.write_dot(db, fp, names, lens, min_n)
        assert read_text(str(tmp_path / "p.tsv")) == pb.getvalue()
        assert read_text(str(tmp_path / "g.dot")) == db.getvalue()
    assert "scaf_num=39" in read_text(str(tmp_path / "g.dot"))


def test_native_pair_writers_report_io_errors(tmp_path):
    """a file that cannot be created or written completely is an OSError that carries the errno (as the Python writers it
    replaces would raise), and no partial .pairs.tsv / .dot is left behind (written beside its place, renamed when complete)."""
    import errno
    t = pairing.PairTally(["a", "b"], np.array([100, 100], np.uint32), 32, 10)
    t.merge((np.array([0], np.uint32), np.array([1], np.uint8), np.array([1], np.uint32), np.array([1], np.uint8), np.array([2], np.uint32),
             np.array([0, 2], np.uint64), np.array([5, 7], np.int64)))
    with pytest.raises(OSError) as exc:
        t.write(1, 1, str(tmp_path / "no_such_dir" / "p.tsv"), str(tmp_path / "g.dot"))
    assert exc.value.errno == errno.ENOENT
    assert os.listdir(tmp_path) == []
    if os.path.exists("/dev/full"):  # every write fails with ENOSPC; the rename of the temporary never happens
        os.symlink("/dev/full", tmp_path / f"g.dot.tmp.{os.getpid()}")
        with pytest.raises(OSError) as exc:
            t.write(1, 1, None, str(tmp_path / "g.dot"))
        assert exc.value.errno == errno.ENOSPC and not os.path.exists(tmp_path / "g.dot")
    assert t.write(1, 1, str(tmp_path / "p.tsv"), str(tmp_path / "g2.dot")) == 1
    assert sorted(f for f in os.listdir(tmp_path) if not f.startswith("g.dot.tmp")) == ["g2.dot", "p.tsv"]


def test_btllib_indexlr_refuses_the_modes_it_does_not_have():
    """flags other than LONG_MODE raise instead of being ignored (no device needed: checked before one is opened)."""
    import ntlink_amd.btllib as btllib
    for flags in (btllib.IndexlrFlag.SHORT_MODE, btllib.IndexlrFlag.LONG_MODE | btllib.IndexlrFlag.BX, 0):
        with pytest.raises(NotImplementedError):
            btllib.Indexlr("x.fa", 20, 10, flags)

"""Parity checks shared by the GPU tests (real library, through the C ABI) and the SIMT-mock tests.
Every check compares the product's records with the oracle's on the same inputs, bit for bit."""
import os

import numpy as np

import oracle
from ntlink_amd import capi
from helpers import REF, contig_ids, load_scenario, parse_indexlr


def offsets_of(seqs):
    off = np.zeros(len(seqs) + 1, np.uint64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    return off


def check_sketch(dev, seqs, k, w, threads=0, info=None):
    """Device sketch == oracle sketch (offsets, hashes, positions, strands).  info (dict): strips / redo_strips of the run."""
    with dev.batch(seqs) as b, dev.sketch(b, k, w) as sk:
        off, h, p, s = sk.download()
        if info is not None:
            info.update(strips=sk.strips, redo_strips=sk.redo_strips, fallback_strips=sk.fallback_strips, from_lists=sk.from_lists)
    ooff, oh, op, os_ = oracle.sketch_batch(b"".join(seqs), offsets_of(seqs), k, w, threads=threads)
    assert np.array_equal(off, ooff), "per-sequence minimizer counts differ"
    assert np.array_equal(p, op), "positions differ"
    assert np.array_equal(h, oh), "hashes differ"
    assert np.array_equal(s, os_), "strands differ"
    return len(h)


def small_window_sequences(seed=21, n_long=6, long_len=40000):
    """Sequences for the small-window pass (sketch_small_kernel, 2 <= w <= 15): sequences of several strips (the wavefronts that lie
    wholly inside a sequence take the path without per-window checks), strip-boundary lengths of ITS strips (4096 elements, 4079 own
    windows), low complexity (equal hashes inside one window: the rightmost wins), N runs across strips."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)

    def rnd(n):
        return bytes(acgt[rng.integers(0, 4, n)])
    out = [rnd(int(long_len + rng.integers(0, 5000))) for _ in range(n_long)]
    out += [rnd(n) for n in (4079, 4080, 4081, 4093, 4094, 4095, 4096, 4097, 4110, 4111, 4112, 8158, 8159, 8160, 8175, 8190, 12240)]
    out += [b"A" * 9000, b"AC" * 4500, b"AAC" * 3000, rnd(3000) + b"T" * 2500 + rnd(3000),
            rnd(5000) + b"N" + rnd(4070) + b"NN" + rnd(13) + b"N" * 30 + rnd(9000) + b"N", b"N" * 17 + rnd(8200) + b"n" * 3 + rnd(40)]
    return out


def check_small_windows(dev, ws, ks=(15, 20, 33), seqs=None, fuzz_seeds=(4, 10)):
    """sketch_small_kernel == oracle for every window size in ws (and the round-1 forms it replaced, NTL_SKETCH_SMALL=0, on one k)."""
    import fuzz_cases
    seqs = small_window_sequences() if seqs is None else seqs
    fz = [q for sd in fuzz_seeds for q in fuzz_cases.fuzz_sequences(sd)]
    n = 0
    for w in ws:
        for k in ks:
            st = {}
            n += check_sketch(dev, seqs + edge_sequences() + fz, k, w, info=st)
            assert st["redo_strips"] == 0 and not st["from_lists"]
    return n


def near_tie_sequences(k, n_pairs, seed=11, pool=1 << 22, flank=30, third=False):
    """Sequences in which two DIFFERENT k-mers whose hashes agree in bits 33..63 (so the window pass's ring keys are within
    one of each other: it cannot order them) are the two smallest k-mers of one window: A + filler + B.  Found by brute force
    over the k-mers of a random sequence; the lowest such pairs, so that random filler rarely goes below them.
    third: a still smaller k-mer Z follows closely (A + B + Z inside one window of 40), so that the later one of the pair is a
    minimizer only if it is the smaller of the two -- an early, unproven bit for it would show."""
    rng = np.random.default_rng(seed)
    text = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), pool))
    h0, _, pos, _ = oracle.hash_seq(text, k)
    c = h0 >> np.uint64(33)
    order = np.argsort(c, kind="stable")
    cs, hs = c[order], h0[order]
    same = np.flatnonzero((cs[1:] == cs[:-1]) & (hs[1:] != hs[:-1]))
    z0 = int(pos[order[0]])
    Z = text[z0:z0 + k]
    fill = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))
    seqs = []
    for i in same:
        if third and cs[i] == cs[0]:
            continue
        a, b = int(pos[order[i]]), int(pos[order[i + 1]])
        A, B = text[a:a + k], text[b:b + k]
        lim = int(min(hs[i], hs[i + 1]))
        for _ in range(300):
            first, second = (A, B) if rng.integers(2) else (B, A)
            if third:
                seq = fill(flank) + first + fill(int(rng.integers(1, 4))) + second + fill(int(rng.integers(1, 4))) + Z + fill(flank)
            else:
                seq = fill(flank) + first + fill(int(rng.integers(3, 12))) + second + fill(flank)
            hh = oracle.hash_seq(seq, k)[0]
            if int(np.sum(hh < np.uint64(lim))) == (1 if third else 0) and int(np.sum(hh >> np.uint64(33) == cs[i])) == 2:
                seqs.append(seq)
                break
        if len(seqs) == n_pairs:
            break
    assert len(seqs) == n_pairs, "not enough near-tie pairs in the pool"
    return seqs


def assert_same_records(got, exp):
    for nm in ("maps", "hits", "pafs"):
        a, b = got[nm], exp[nm]
        assert len(a) == len(b), f"{nm}: {len(a)} records, oracle has {len(b)}"
        if a.tobytes() != b.tobytes():
            av = a.view(np.uint8).reshape(len(a), -1)
            bv = b.view(np.uint8).reshape(len(b), -1)
            i = int(np.flatnonzero((av != bv).any(axis=1))[0])
            raise AssertionError(f"{nm}[{i}]: {a[i]} != oracle {b[i]}")


def check_pair_arrays(dev, coff, ch, cp, cs, ctg_len, roff, rlen, rh, rp, rs, **kw):
    """index + map on given sketches (device) == oracle."""
    with dev.sketch_from_arrays(coff, ch, cp, cs) as csk, dev.index(csk, ctg_len) as ix, \
            dev.sketch_from_arrays(roff, rh, rp, rs) as rsk, dev.map(ix, rsk, rlen, **kw) as res:
        got = res.download()
        nix = len(ix)
        nhit = res.n_index_hits
    cid = contig_ids(coff) if len(ch) else np.empty(0, np.uint32)
    oix = oracle.Index(ch, cid, cp, cs)
    exp = oracle.map_reads(oix, ctg_len, roff, rlen, rh, rp, rs, threads=0, **kw)
    assert nix == len(oix), "index size differs"
    assert_same_records(got, exp)
    return got, nhit


def scenario_arrays(name):
    meta, ctext, rtext, exp = load_scenario(name)
    p, k = meta["params"], meta["k"]
    cn, _, coff, ch, cp, cs = parse_indexlr(ctext, False)
    ids = [meta["ctg_names"].index(n) for n in cn]
    nctg = len(meta["ctg_names"])
    cnt = np.zeros(nctg, np.uint64)
    cnt[ids] = np.diff(coff)
    full_off = np.zeros(nctg + 1, np.uint64)
    np.cumsum(cnt, out=full_off[1:])
    rn, rlen, roff, rh, rp, rs = parse_indexlr(rtext, True)
    kw = dict(k=k, z=p.get("z", 1000), x=p.get("x", 0.0), sensitive=p.get("sensitive", False),
              repeat_filter=p.get("repeat_filter", False))
    return meta, exp, (full_off, ch, cp, cs, np.array(meta["ctg_len"], np.uint32), roff, rlen, rh, rp, rs), kw, rn


def check_scenario(dev, name):
    _, _, arrs, kw, _ = scenario_arrays(name)
    return check_pair_arrays(dev, *arrs, **kw)


def check_full_pipeline(dev, contigs, reads, k, w, **kw):
    """FASTA-level: device sketch of contigs and reads -> index -> map, everything against the oracle."""
    ctg_len = np.array([len(s) for s in contigs], np.uint32)
    rlen = np.array([len(s) for s in reads], np.uint32)
    with dev.batch(contigs) as cb, dev.sketch(cb, k, w) as csk, dev.index(csk, ctg_len) as ix, \
            dev.batch(reads) as rb, dev.sketch(rb, k, w) as rsk, dev.map(ix, rsk, rlen, k=k, **kw) as res:
        got = res.download()
        coff, ch, cp, cs = csk.download()
        roff, rh, rp, rs = rsk.download()
        nix = len(ix)
    ooff, oh, op, os_ = oracle.sketch_batch(b"".join(contigs), offsets_of(contigs), k, w)
    assert np.array_equal(coff, ooff) and np.array_equal(ch, oh) and np.array_equal(cp, op) and np.array_equal(cs, os_)
    qoff, qh, qp, qs = oracle.sketch_batch(b"".join(reads), offsets_of(reads), k, w)
    assert np.array_equal(roff, qoff) and np.array_equal(rh, qh) and np.array_equal(rp, qp) and np.array_equal(rs, qs)
    oix = oracle.Index(oh, contig_ids(ooff), op, os_)
    exp = oracle.map_reads(oix, ctg_len, qoff, rlen, qh, qp, qs, k=k, threads=0, **kw)
    assert nix == len(oix)
    assert_same_records(got, exp)
    return got


def check_small_window_pipeline(dev, k, w, **kw):
    """The whole path at a small window -- dense sketches (a minimizer every (w + 1) / 2 bases): sketch_small_kernel on both sides, the
    index, and the mapping from all three forms of the read sketch (records, made for the index, record-less) against the oracle."""
    from helpers import REF
    contigs = fixture_seqs("scaffolds_4.fa")
    reads = fixture_seqs("long_reads_4_top5.fa")
    got = check_full_pipeline(dev, contigs, reads, k, w, **kw)
    ctg_len = np.array([len(s) for s in contigs], np.uint32)
    rlen = np.array([len(s) for s in reads], np.uint32)
    for records in (True, False):
        with dev.batch(contigs) as cb, dev.sketch(cb, k, w) as csk, dev.index(csk, ctg_len) as ix, dev.batch(reads) as rb, \
                dev.sketch(rb, k, w, index=ix, records=records) as rsk, dev.map(ix, rsk, rlen, k=k, **kw) as res:
            assert_same_records(res.download(), got)
    return got


def check_handles_outlive_their_inputs(dev, contigs, reads, k, w, n_live=700, tiny_len=None, **kw):
    """Two promises of the header's "Asynchrony" paragraph: (1) completed handles cost no page-locked slot -- more
    than the context's 512 slots' worth of completed sketches and map results stay alive side by side; (2) an index may be
    destroyed as soon as the calls that took it have returned -- also when the read sketch then turns out to have overflowed its
    record array and both it and the mapping are made AGAIN from that index when the result is finally asked for (the caller
    forces that with NTL_SKETCH_CAP_GUESS)."""
    ctg_len = np.array([len(s) for s in contigs], np.uint32)
    rlen = np.array([len(s) for s in reads], np.uint32)
    ooff, oh, op, os_ = oracle.sketch_batch(b"".join(contigs), offsets_of(contigs), k, w)
    oix = oracle.Index(oh, contig_ids(ooff), op, os_)
    qoff, qh, qp, qs = oracle.sketch_batch(b"".join(reads), offsets_of(reads), k, w)
    exp = oracle.map_reads(oix, ctg_len, qoff, rlen, qh, qp, qs, k=k, threads=0, **kw)
    # (2) first: index, contig sketch and read batch all gone before anything is asked
    cb = dev.batch(contigs)
    csk = dev.sketch(cb, k, w)
    ix = dev.index(csk, ctg_len)
    rb = dev.batch(reads)
    rsk = dev.sketch(rb, k, w, index=ix, records=False)  # as the fused driver makes it: for this map only
    res = dev.map(ix, rsk, rlen, k=k, **kw)
    ix.close(); csk.close(); cb.close(); rb.close()
    junk = [dev.batch([b"ACGT" * 300]) for _ in range(8)]  # whatever the freed blocks are handed out for next
    got = res.download()
    assert_same_records(got, exp)
    assert rsk.count == len(qh)
    res.close(); rsk.close()
    for j in junk:
        j.close()
    # (1) n_live completed sketches + map results alive at once
    with dev.batch(contigs) as cb, dev.sketch(cb, k, w) as csk, dev.index(csk, ctg_len) as ix:
        tiny = [reads[0][:tiny_len]] if tiny_len else reads[:1]
        tl = np.array([len(tiny[0])], np.uint32)
        toff, th, tp, ts = oracle.sketch_batch(tiny[0], offsets_of(tiny), k, w)
        live = []
        with dev.batch(tiny) as tb:
            for i in range(n_live):
                sk = dev.sketch(tb, k, w, index=ix)
                mr = dev.map(ix, sk, tl, k=k, **kw)
                assert sk.count == len(th) and mr.counts()[0] >= 0  # completes both: their slots and events go back here
                live.append((sk, mr))
        first = live[0][1].download()
        last = live[-1][1].download()
        assert_same_records(first, last)
        for sk, mr in live:
            mr.close(); sk.close()
    dev.sync()
    return len(got["maps"])


def check_device_text(dev, contigs, reads, k, w, read_names=None, ctg_names=None, **kw):
    """The lines of .verbose_mapping.tsv and .paf made on the device (ntl_mapres_format) == the host emitters' bytes for the same
    records (ntl_write_verbose / ntl_write_paf) == the oracle's formatting of the oracle's records; the mappings' first and last
    hits == the records'; with only one of the two texts asked for, the other one is empty and the ends still come."""
    import tempfile
    from ntlink_amd import formats
    ctg_len = np.array([len(s) for s in contigs], np.uint32)
    rlen = np.array([len(s) for s in reads], np.uint32)
    read_names = read_names or [f"read_{i}/x{'y' * (i % 7)}" for i in range(len(reads))]
    ctg_names = ctg_names or [f"ctg{i:05d}" for i in range(len(contigs))]
    with dev.batch(contigs) as cb, dev.sketch(cb, k, w) as csk, dev.index(csk, ctg_len) as ix, dev.batch(reads) as rb, \
            dev.sketch(rb, k, w, index=ix) as rsk, dev.map(ix, rsk, rlen, k=k, **kw) as res, \
            dev.names(read_names, rlen) as rn, dev.names(ctg_names, ctg_len) as cn:
        rec = res.download()
        with res.format(rn, cn, True, True) as txt:
            both = txt.download()
            got_v, got_p = bytes(both["verbose"]), bytes(both["paf"])
            maps, ends = both["maps"].copy(), both["ends"].copy()
            dev.pinned_release(both["_pinned"])
        with res.format(rn, cn, False, True) as txt:
            only_p = txt.download()
            assert len(only_p["verbose"]) == 0 and bytes(only_p["paf"]) == got_p and np.array_equal(only_p["ends"], ends)
            dev.pinned_release(only_p["_pinned"])
        with res.format(rn, cn, True, False) as txt:
            only_v = txt.download()
            assert len(only_v["paf"]) == 0 and bytes(only_v["verbose"]) == got_v
            dev.pinned_release(only_v["_pinned"])
    with tempfile.TemporaryFile("w+") as fv, tempfile.TemporaryFile("w+") as fp:
        formats.write_verbose(fv, rec, read_names, ctg_names)
        formats.write_paf(fp, rec, read_names, rlen, ctg_names, ctg_len)
        fv.flush(); fp.flush(); fv.seek(0); fp.seek(0)
        exp_v, exp_p = fv.read().encode(), fp.read().encode()
    assert got_v == exp_v, (len(got_v), len(exp_v))
    assert got_p == exp_p, (len(got_p), len(exp_p))
    assert got_p.decode() == oracle.format_paf(rec, read_names, rlen, ctg_names, ctg_len)
    assert np.array_equal(maps, rec["maps"])
    m = rec["maps"]
    if len(m):
        first = rec["hits"][m["hit_off"].astype(np.int64)]
        last = rec["hits"][(m["hit_off"] + m["n_hits"] - 1).astype(np.int64)]
        assert np.array_equal(ends[0::2], first) and np.array_equal(ends[1::2], last)
    return len(m), len(got_v), len(got_p)


def fixture_seqs(fname):
    return [s for _, s in oracle.read_fastx(os.path.join(REF, fname))]


def tiny_sequences(n=3000, seed=8):
    """Thousands of sequences of 0..120 bases: more than 512 sequence starts inside one 65536-position
    tile of the emit kernel (uncached sequence lookup) and several rounds of its workgroup-wide search."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGTN", np.uint8)
    return [bytes(acgt[rng.choice(5, int(rng.integers(0, 121)), p=[0.245, 0.245, 0.245, 0.245, 0.02])]) for _ in range(n)]


def edge_sequences(seed=5):
    rng = np.random.default_rng(seed)

    def rnd(n):
        return bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)])
    return [b"", b"ACGT", rnd(50), b"A" * 3000, b"AC" * 2000, rnd(31), rnd(32), rnd(131), rnd(132), b"N" * 500,
            rnd(700) + b"N" * 40 + rnd(800) + b"n" + rnd(33) + b"NN" + rnd(5000) + b"RYK" + rnd(20),
            b"N" + rnd(4500), rnd(4500) + b"N", rnd(4085), rnd(4086), rnd(4087), rnd(8200)]


def check_anchor_cases(dev, max_cases=None):
    """ntlink_amd.anchor.get_accepted_anchor_contigs == the reference's function on the golden cases
    (tests/golden/gen/anchor_cases.json, made by tests/golden/gen_goldens_anchor.py from the imported reference)."""
    import argparse
    import gzip
    import json
    import os
    from collections import namedtuple
    from helpers import GEN
    from ntlink_amd import anchor
    Scaffold = namedtuple("Scaffold", ["id", "length"])
    cases = json.load(open(os.path.join(GEN, "anchor_cases.json")))
    if max_cases:
        cases = cases[::max(1, len(cases) // max_cases)]
    cache = {}
    n = 0
    for c in cases:
        name = c["scenario"]
        if name not in cache:
            d = os.path.join(GEN, "synthetic")
            meta = json.load(open(os.path.join(d, name + ".json")))
            p = meta["params"]
            args = argparse.Namespace(k=meta["k"], z=p.get("z", 1000), x=p.get("x", 0.0), sensitive=p.get("sensitive", False))
            scaffolds = {nm: Scaffold(nm, ln) for nm, ln in zip(meta["ctg_names"], meta["ctg_len"])}
            mx_info, dup = {}, set()
            for line in gzip.open(os.path.join(d, name + ".contigs.tsv.gz"), "rt"):
                f = line.strip().split("\t")
                if len(f) > 1:
                    for tok in f[1].split(" "):
                        mx, pos, strand = tok.split(":")
                        if mx in mx_info:
                            dup.add(mx)
                        else:
                            mx_info[mx] = anchor.Minimizer(f[0], int(pos), strand)
            mx_info = {m: v for m, v in mx_info.items() if m not in dup}
            reads = [line.strip().split("\t") for line in gzip.open(os.path.join(d, name + ".reads.tsv.gz"), "rt")]
            cache[name] = (args, scaffolds, mx_info, reads, anchor.AnchorMapper(mx_info, scaffolds, dev))
        args, scaffolds, mx_info, reads, mapper = cache[name]
        f = reads[c["read_index"]]
        assert f[0] == c["read"]
        mx_list = [(mx, int(pos), strand) for mx, pos, strand in (t.split(":") for t in f[2].split(" ")) if mx in mx_info]
        for acc, order in (mapper.map_many([(mx_list, int(f[1]))], args)[0],
                           anchor.get_accepted_anchor_contigs(mx_list, int(f[1]), scaffolds, mx_info, args, dev=dev) if n % 10 == 0 else (None, None)):
            if acc is None:
                continue
            assert order == c["order"], (name, c["read"])
            for ctg in order:
                got = [[h.mx, h.ctg_pos, h.ctg_strand, h.read_pos, h.read_strand] for h in acc[ctg].hits]
                assert got == c["hits"][ctg], (name, c["read"], ctg)
                assert acc[ctg].hit_count == len(got) and acc[ctg].contig == ctg
        n += 1
    for _a, _s, _m, _r, mapper in cache.values():
        mapper.close()
    return n


# ---- row f3: overlap-stage consumer ---------------------------------------------------------------

def overlap_cases():
    import json
    from helpers import GEN
    return json.load(open(os.path.join(GEN, "overlap", "cases.json")))


def check_overlap_case(dev, case, tmp_path):
    """ntlink_amd.overlap (native `H:pos` parser + ovl_* kernels) == the imported reference's read_minimizers /
    read_minimizers_path on the same TSV and valid regions (tests/golden/gen_goldens_overlap.py)."""
    import gzip
    import json
    from helpers import GEN
    from ntlink_amd import overlap
    d = os.path.join(GEN, "overlap")
    c = json.load(gzip.open(os.path.join(d, case + ".json.gz"), "rt"))
    valid = {n: [tuple(r) for r in v] for n, v in c["valid"].items()}
    text = gzip.open(os.path.join(d, c["tsv"]), "rt").read()
    tsv = os.path.join(str(tmp_path), "in.tsv")
    open(tsv, "w").write(text)
    mx_info, mxs = overlap.read_minimizers(tsv, valid, dev=dev)
    assert set(mxs) == set(c["expected"]) and set(k for k in mx_info if k in mxs) == set(mxs)
    n_kept = 0
    for n, e in c["expected"].items():
        assert mxs[n] == [e["mxs"]], n
        assert list(mx_info[n]) == e["mxs"] and [mx_info[n][m] for m in e["mxs"]] == [(n, p) for p in e["pos"]], n
        n_kept += len(e["mxs"])
    # the path form: one call per LAST marker
    lines = text.splitlines(keepends=True)
    marked = []
    for i, line in enumerate(lines):
        marked.append(line)
        if i % 3 == 2:
            marked.append("LASTntLink_%d\t\n" % (i // 3))
    reader = iter(marked)
    for chunk in c["path_chunks"]:
        _info, got = overlap.read_minimizers_path(reader, valid, dev=dev)
        assert {n: m[0] for n, m in got.items()} == chunk
    assert next(reader, None) is None
    return n_kept


def check_overlap_random(dev, seed, nseq=40, max_len=30000, k=15, w=5):
    """Device sketch at the overlap stage's density -> ntl_overlap_filter == the oracle's restatement; sequences with tandem
    repeats so that duplicated hashes are common, random region lists (none / empty / overlapping / beyond the end)."""
    rng = np.random.default_rng(seed)
    seqs, regions = [], []
    for i in range(nseq):
        n = int(rng.integers(1, max_len))
        s = rng.integers(0, 4, n).astype(np.uint8)
        if i % 3 == 0 and n > 200:
            unit = int(rng.integers(1, 60)); a = int(rng.integers(0, n - 100)); m = int(rng.integers(50, n - a))
            s[a:a + m] = np.resize(s[a:a + unit], m)
        seqs.append(bytes(np.frombuffer(b"ACGT", np.uint8)[s]))
        r = rng.random()
        if r < 0.2:
            regions.append(None)
        elif r < 0.3:
            regions.append([])
        else:
            regions.append([tuple(sorted(int(x) for x in rng.integers(0, n + 100, 2))) for _ in range(int(rng.integers(1, 4)))])
    off = np.zeros(nseq + 1, np.uint64)
    starts, ends = [], []
    for i, reg in enumerate(regions):
        for a, b in reg or ():
            starts.append(a); ends.append(b)
        off[i + 1] = len(starts)
    with dev.batch(seqs) as b, dev.sketch(b, k, w) as sk:
        m_off, h, p, _ = sk.download()
        with dev.overlap_filter(sk, off, np.array(starts, np.uint32), np.array(ends, np.uint32)) as kept:
            g_off, gh, gp, _gs = kept.download()
    e_off, eh, ep = oracle.overlap_filter(m_off, h, p, regions)
    assert np.array_equal(g_off, e_off) and np.array_equal(gh, eh) and np.array_equal(gp, ep)
    assert 0 < len(eh) < len(h)
    return len(eh), len(h)


def check_probe_forms(dev, contigs, reads, k, w, **kw):
    """Both forms of the index lookup on one index: the first batch goes through the slot tags, a batch after one that found
    most of its minimizers reads the slots directly (probe_kernel<false>); both must equal the oracle."""
    ctg_len = np.array([len(s) for s in contigs], np.uint32)
    with dev.batch(contigs) as cb, dev.sketch(cb, k, w) as csk, dev.index(csk, ctg_len) as ix:
        coff, ch, cp, cs = csk.download()
        oix = oracle.Index(ch, contig_ids(coff), cp, cs)
        fractions = []
        for rs in (reads, reads, list(reversed(reads))):
            rlen = np.array([len(s) for s in rs], np.uint32)
            with dev.batch(rs) as rb, dev.sketch(rb, k, w) as rsk, dev.map(ix, rsk, rlen, k=k, **kw) as res:
                got = res.download()
                roff, rh, rp, rstr = rsk.download()
                fractions.append(res.n_index_hits / max(rsk.count, 1))
                # the sketch made for the index (lookups inside emit_kernel, no probe pass): same minimizers, same records
                with dev.sketch(rb, k, w, index=ix) as isk, dev.map(ix, isk, rlen, k=k, **kw) as ires:
                    got2 = ires.download()
                    ioff, ih, ip, istr = isk.download()
                    assert ires.n_index_hits == res.n_index_hits
                assert np.array_equal(ioff, roff) and np.array_equal(ih, rh) and np.array_equal(ip, rp) and np.array_equal(istr, rstr)
                # ... and the one made for this map only (ntl_sketch_run_for_map: positions and strands without the records)
                with dev.sketch(rb, k, w, index=ix, records=False) as lsk, dev.map(ix, lsk, rlen, k=k, **kw) as lres:
                    got3 = lres.download()
                    assert lres.n_index_hits == res.n_index_hits and lsk.count == rsk.count
                    for bad in (lambda: lsk.download(), lambda: dev.index(lsk, rlen)):
                        try:
                            bad()
                        except capi.NtlError as exc:
                            assert exc.code == capi.NTL_EINVAL
                        else:
                            raise AssertionError("a sketch without records handed out records")
                    with dev.index(csk, ctg_len) as other:  # another index, even of the same contigs: not the one it was made for
                        try:
                            dev.map(other, lsk, rlen, k=k, **kw)
                        except capi.NtlError as exc:
                            assert exc.code == capi.NTL_EINVAL
                        else:
                            raise AssertionError("a sketch without records was mapped against another index")
            exp = oracle.map_reads(oix, ctg_len, roff, rlen, rh, rp, rstr, threads=0, k=k, **kw)
            assert_same_records(got, exp)
            assert_same_records(got2, exp)
            assert_same_records(got3, exp)
    return fractions


def check_async_order(dev, contigs, reads, k, w, **kw):
    """The calls of the hot path only queue device work: batches are sketched for the index and mapped back to back, their
    inputs destroyed at once, the NEXT batch queued before the previous one's records are asked for -- records == oracle, per
    batch.  Then a batch nobody ever asks about: destroying its handles is allowed, and a clean run leaves ntl_ctx_sync clean."""
    ctg_len = np.array([len(s) for s in contigs], np.uint32)
    csk = dev.sketch(dev.batch(contigs), k, w)        # the batch handle is dropped at once: the sketch keeps what it needs
    ix = dev.index(csk, ctg_len)
    ooff, oh, op, os_ = oracle.sketch_batch(b"".join(contigs), offsets_of(contigs), k, w)
    oix = oracle.Index(oh, contig_ids(ooff), op, os_)
    groups = [reads[i::3] for i in range(3)]
    held = []
    n_maps = 0
    for g in groups + [None]:
        if g is not None:
            rl = np.array([len(s) for s in g], np.uint32)
            rb = dev.batch(g)
            rsk = dev.sketch(rb, k, w, index=ix)
            res = dev.map(ix, rsk, rl, k=k, **kw)
            rb.close()
            rsk.close()                                # before anybody asked for its count
            held.append((g, rl, res))
        if len(held) > 1 or (g is None and held):
            gg, rl, res = held.pop(0)
            got = res.download()
            res.close()
            qoff, qh, qp, qs = oracle.sketch_batch(b"".join(gg), offsets_of(gg), k, w)
            exp = oracle.map_reads(oix, ctg_len, qoff, rl, qh, qp, qs, k=k, threads=0, **kw)
            assert_same_records(got, exp)
            n_maps += len(got["maps"])
    rb = dev.batch(reads)
    rsk = dev.sketch(rb, k, w, index=ix)
    res = dev.map(ix, rsk, np.array([len(s) for s in reads], np.uint32), k=k, **kw)
    res.close(); rsk.close(); rb.close()               # never looked at
    dev.sync()                                         # nothing failed behind our back
    ix.close(); csk.close()
    return n_maps


def check_strip_lists(dev, monkeypatch, scale=1):
    """Round 5: for the windows ntLink runs with the window passes write per-strip minimizer LISTS (every minimizer by the strip that
    owns the first window it is the minimum of) instead of a bitmask.  Same records as the oracle whichever pass lists a strip --
    sketch_wave_kernel, the block-minima pass for what it gives up, the exact pass (forced: every strip through all three), the
    multi-run pass (N in the sequences) -- with strips across several strips' seams (sequences of many strips), with lists that
    live in the pool (slot of one entry), on low-complexity sequence (a minimizer per base), and when the pool runs out (the sketch
    is made again through the bitmask)."""
    import fuzz_cases
    rng = np.random.default_rng(15)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seqs = [bytes(acgt[rng.integers(0, 4, n)]) for n in tuple(n * scale for n in (30000, 12000, 4127, 300, 8000, 281, 3871, 0, 31, 9000))]
    withn = [s[:2000] + b"N" * 7 + s[2000:5000] + b"NN" + s[5000:] for s in seqs[:2]] + seqs[2:5]
    lowc = [b"A" * 6000 + seqs[1][:3000] + b"ACACACAC" * 700 + seqs[4], b"T" * 300, seqs[0][:9000]]
    for k, w in ((32, 250), (24, 100), (40, 137)):
        info = {}
        assert check_sketch(dev, seqs, k, w, info=info) > 0
        assert info["from_lists"], (k, w, info)
        assert check_sketch(dev, withn, k, w, info=info) > 0 and info["from_lists"]
        assert check_sketch(dev, lowc, k, w, info=info) > 0 and info["from_lists"] and info["redo_strips"] > 0
    assert check_sketch(dev, fuzz_cases.fuzz_sequences(2)[:12], 32, 250, info=info) > 0 and info["from_lists"]
    with monkeypatch.context() as m:
        m.setenv("NTL_LIST_SLOT", "1")  # every list of more than one entry lives in the pool
        assert check_sketch(dev, seqs + lowc, 32, 250, info=info) > 0 and info["from_lists"]
        m.setenv("NTL_LIST_POOL", "500")  # ... which runs out
        assert check_sketch(dev, seqs + lowc, 32, 250, info=info) > 0 and not info["from_lists"]
        check_full_pipeline(dev, fixture_seqs("scaffolds_4.fa"), fixture_seqs("long_reads_4_top5.fa"), 40, 100, z=1000)
    with monkeypatch.context() as m:
        m.setenv("NTL_SKETCH_FORCE_REDO", "1")
        assert check_sketch(dev, seqs + withn, 32, 250, info=info) > 0 and info["from_lists"]
        assert info["fallback_strips"] == info["redo_strips"] > 0
    with monkeypatch.context() as m:
        m.setenv("NTL_SKETCH_THRESH", "4")  # most strips come back from the block-minima pass
        assert check_sketch(dev, seqs, 32, 250, info=info) > 0 and info["from_lists"] and info["fallback_strips"] > 3
    with monkeypatch.context() as m:
        m.setenv("NTL_SKETCH_LISTS", "0")
        assert check_sketch(dev, seqs, 32, 250, info=info) > 0 and not info["from_lists"]
    assert check_sketch(dev, seqs, 32, 64, info=info) > 0 and not info["from_lists"]  # (a window of the bitmask passes)

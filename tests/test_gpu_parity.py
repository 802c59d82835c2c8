"""Parity tests proper: the hipcc-built library on a real MI355X, called through the C ABI
(ntlink_amd.capi -> libntlink_hip.so), against the oracle and the committed golden fixtures."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
import parity_cases as pc
from helpers import FIXTURES, GEN, REF, SCENARIOS, TEST7_PAF, read_text
from ntlink_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    d = capi.Device(0)          # raises if the HIP extension or the GPU is missing: no fallback
    assert "gfx950" in d.name, d.name
    yield d
    d.close()


@pytest.mark.parametrize("tag,target,reads,k,w,gold", [f for f in FIXTURES if f[5]])
def test_contig_sketch_equals_reference_golden_tsv(dev, tag, target, reads, k, w, gold):
    recs = list(oracle.read_fastx(os.path.join(REF, target)))
    with dev.batch([s for _, s in recs]) as b, dev.sketch(b, k, w) as sk:
        off, h, p, s = sk.download()
    text = oracle.format_indexlr([(n, len(q), h[int(off[i]):int(off[i + 1])], p[int(off[i]):int(off[i + 1])],
                                   s[int(off[i]):int(off[i + 1])]) for i, (n, q) in enumerate(recs)])
    assert text == read_text(os.path.join(REF, "expected_outputs", gold + ".tsv"))


@pytest.mark.parametrize("tag,target,reads,k,w,gold", FIXTURES)
def test_read_sketch_md5(dev, tag, target, reads, k, w, gold):
    recs = list(oracle.read_fastx(os.path.join(REF, reads)))
    with dev.batch([s for _, s in recs]) as b, dev.sketch(b, k, w) as sk:
        off, h, p, s = sk.download()
    text = oracle.format_indexlr([(n, len(q), h[int(off[i]):int(off[i + 1])], p[int(off[i]):int(off[i + 1])],
                                   s[int(off[i]):int(off[i + 1])]) for i, (n, q) in enumerate(recs)], with_len=True)
    summ = json.load(open(os.path.join(GEN, "fixtures", "summary.json")))[tag]
    assert hashlib.md5(text.encode()).hexdigest() == summ["read_tsv_md5"]


@pytest.mark.parametrize("k,w", [(32, 100), (32, 250), (24, 100), (40, 100), (15, 5), (20, 10), (8, 4), (5, 17), (3, 1), (64, 1000)])
def test_sketch_edge_cases(dev, k, w):
    pc.check_sketch(dev, pc.edge_sequences(), k, w)


@pytest.mark.parametrize("ws", [(2, 3, 4), (5, 6, 7), (8, 9, 10), (11, 12, 13), (14, 15)], ids=lambda t: "w" + "_".join(map(str, t)))
def test_small_window_pass(dev, ws, monkeypatch):
    """Round 6: windows of 2 .. 15 k-mers -- the overlap stage's k15 w5 (ntLink:243-251) and the gap filler's
    k20 w10 (bin/ntlink_patch_gaps.py:417-441) among them -- go through sketch_small_kernel<W> (16 k-mers per lane, window minima
    in registers): every window size against the oracle on multi-strip sequences, its strips' boundary lengths, low complexity and
    N runs; and the forms it replaced (four / one k-mer per lane) still agree."""
    assert pc.check_small_windows(dev, ws) > 0
    monkeypatch.setenv("NTL_SKETCH_SMALL", "0")
    pc.check_small_windows(dev, ws[:1], ks=(20,))


@pytest.mark.parametrize("k,w,flags", [(15, 5, {}), (32, 8, {}), (20, 10, {"sensitive": True}), (24, 3, {"repeat_filter": True})], ids=["k15w5", "k32w8", "k20w10s", "k24w3r"])
def test_small_window_whole_path(dev, k, w, flags):
    """Dense sketches through the whole path (sketch_small_kernel on contigs and reads, index, the three forms of the read sketch,
    map: thousands of hits per read, the large staging classes and the overflow kernel) == oracle."""
    got = pc.check_small_window_pipeline(dev, k, w, z=1000, **flags)
    assert (len(got["maps"]) > 0) == (k != 15)  # (k15 w5 on this fixture: every read's hits are filtered out, on both sides)


@pytest.mark.parametrize("k,w", [(12, 8), (20, 40), (5, 1)])
def test_sketch_many_tiny_sequences(dev, k, w):
    """> 512 sequence starts per emit tile and several rounds of the workgroup-wide sequence search."""
    pc.check_sketch(dev, pc.tiny_sequences(20000), k, w)


def test_sketch_empty_batch(dev):
    with dev.batch([]) as b, dev.sketch(b, 32, 100) as sk:
        assert sk.count == 0
    with dev.batch([b"", b"", b"ACGT"]) as b, dev.sketch(b, 32, 100) as sk:
        off, h, p, s = sk.download()
        assert sk.count == 0 and list(off) == [0, 0, 0, 0]


@pytest.mark.parametrize("flags", [{}, {"sensitive": True}, {"repeat_filter": True}], ids=["default", "sensitive", "repeat"])
@pytest.mark.parametrize("tag,target,reads,k,w,gold", FIXTURES)
def test_fixture_pair_outputs(dev, tag, target, reads, k, w, gold, flags):
    """FASTA in -> verbose/PAF/pairs text identical to the vectors made by the imported reference."""
    crecs = list(oracle.read_fastx(os.path.join(REF, target)))
    rrecs = list(oracle.read_fastx(os.path.join(REF, reads)))
    got = pc.check_full_pipeline(dev, [s for _, s in crecs], [s for _, s in rrecs], k, w, z=1000, **flags)
    cn, rn = [n for n, _ in crecs], [n for n, _ in rrecs]
    cl = np.array([len(s) for _, s in crecs], np.uint32)
    rl = np.array([len(s) for _, s in rrecs], np.uint32)
    full = tag + "".join("." + f for f in flags)
    d = os.path.join(GEN, "fixtures")
    assert oracle.format_verbose(got, rn, cn) == read_text(os.path.join(d, full + ".verbose_mapping.tsv"))
    paf = oracle.format_paf(got, rn, rl, cn, cl)
    assert paf == read_text(os.path.join(d, full + ".paf"))
    pairs = oracle.filter_pairs(oracle.tally_pairs(got, rl, cn, cl, k, 10), dict(zip(cn, map(int, cl))), 1)
    assert oracle.format_pairs(pairs) == read_text(os.path.join(d, full + ".pairs.tsv"))
    if tag.startswith("t7") and not flags:
        assert set(paf.splitlines()) == TEST7_PAF


@pytest.mark.parametrize("name", SCENARIOS)
def test_synthetic_scenarios(dev, name):
    """Branches the fixtures never reach (x != 0, z, noisy contigs, subsumption, PAF filter/break)."""
    meta, exp, arrs, kw, rn = pc.scenario_arrays(name)
    got, _ = pc.check_pair_arrays(dev, *arrs, **kw)
    assert oracle.format_verbose(got, rn, meta["ctg_names"]) == exp[".verbose_mapping.tsv"]
    assert oracle.format_paf(got, rn, arrs[6], meta["ctg_names"], arrs[4]) == exp[".paf"]


@pytest.mark.parametrize("cfg", [dict(k=32, w=100, rl=10000, sens=False, err=(0.02, 0.015, 0.015)),
                                 dict(k=32, w=250, rl=15000, sens=False, err=(0.02, 0.015, 0.015)),
                                 dict(k=24, w=100, rl=20000, sens=True, err=(0.001, 0.0005, 0.0005))],
                         ids=["C2like", "C3like", "C5like"])
def test_synthetic_workloads_vs_oracle(dev, cfg):
    """Scaled-down BASELINE configs (genome 3 Mbp, ~6 Mbases of reads), full pipeline, bit-exact."""
    chroms, cbuf, coff, names, _ = synth.make_assembly(1, 1, 12, 250_000, n_run_every=5)
    rbuf, roff, _ = synth.make_reads(2, chroms, 6_000_000, cfg["rl"], *cfg["err"], lognormal_sigma=0.4)
    contigs = [cbuf[int(coff[i]):int(coff[i + 1])].tobytes() for i in range(len(coff) - 1)]
    reads = [rbuf[int(roff[i]):int(roff[i + 1])].tobytes() for i in range(len(roff) - 1)]
    got = pc.check_full_pipeline(dev, contigs, reads, cfg["k"], cfg["w"], z=1000, sensitive=cfg["sens"])
    assert len(got["maps"]) > 0.5 * len(reads)


def test_long_read_uses_global_scratch(dev):
    """A read with far more than 1024 hits and 128 runs (the largest LDS staging of the map kernels), one with few hits on more
    than 128 contigs, and reads of the two larger LDS size classes."""
    rng = np.random.default_rng(3)
    contigs = [bytes(synth.random_bases(rng, 3000)) for _ in range(400)]
    order = rng.permutation(400)
    read = b"".join(contigs[i] for i in order[:300])
    # few hits on many contigs: the hits fit the LDS staging, the runs (> 128) do not -> map_overflow_kernel as well
    patchy = b"".join(contigs[i][1000:1048] for i in order[:200])
    reads = [read, contigs[5] + contigs[5], b"ACGT", patchy, contigs[7][:500], contigs[9] + contigs[11] + contigs[13][:2000], contigs[21] + contigs[22][:1500]]
    got = pc.check_full_pipeline(dev, contigs, reads, 24, 20, z=1000)
    assert 128 < int((got["maps"]["read"] == 3).sum()) and int(got["maps"]["n_hits"][got["maps"]["read"] == 3].sum()) <= 1024
    assert 512 < int(got["maps"]["n_hits"][got["maps"]["read"] == 5].sum()) <= 1024 and 256 < int(got["maps"]["n_hits"][got["maps"]["read"] == 6].sum()) <= 512
    pc.check_full_pipeline(dev, contigs, reads, 24, 20, z=1000, sensitive=True)


def test_full_size_properties(dev):
    """BASELINE-size properties that need no oracle: per-sequence positions strictly increase,
    density ~ 2/(w+1), sketch of a batch == sketches of its halves, mapping is independent of batching."""
    chroms, cbuf, coff, names, _ = synth.make_assembly(7, 1, 20, 500_000)
    k, w = 32, 100
    with dev.batch(cbuf, coff) as b, dev.sketch(b, k, w) as sk:
        off, h, p, s = sk.download()
    n = len(h)
    dens = n / float(coff[-1])
    assert abs(dens - 2.0 / (w + 1)) < 0.002
    for i in range(len(off) - 1):
        q = p[int(off[i]):int(off[i + 1])].astype(np.int64)
        assert np.all(np.diff(q) > 0)
    half = len(coff) // 2
    with dev.batch(cbuf[:int(coff[half])], coff[:half + 1]) as b1, dev.sketch(b1, k, w) as s1:
        o1, h1, p1, st1 = s1.download()
    assert np.array_equal(h[:len(h1)], h1) and np.array_equal(p[:len(p1)], p1) and np.array_equal(s[:len(st1)], st1)


@pytest.mark.parametrize("seed,k,w", [(1, 32, 100), (2, 32, 250), (3, 24, 100), (4, 15, 5), (5, 20, 10), (6, 7, 33), (7, 40, 16),
                                      (8, 80, 100), (9, 100, 47), (10, 12, 2), (11, 31, 64), (12, 64, 129)])
def test_fuzz_sketch(dev, seed, k, w):
    """Adversarial sequences (ties, N patterns, boundary lengths) incl. k > 64 (loop form of the hash init)."""
    import fuzz_cases
    pc.check_sketch(dev, fuzz_cases.fuzz_sequences(seed), k, w)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_mapping(dev, seed):
    import fuzz_cases
    arrs = fuzz_cases.fuzz_mapping(seed)
    kw = dict(k=24, z=[1000, 500, 1000, 1][seed % 4], x=[0.0, 0.0, 1.2, 0.4][seed % 4], sensitive=bool(seed & 1),
              repeat_filter=bool(seed & 2))
    got, _ = pc.check_pair_arrays(dev, *arrs, **kw)
    assert len(got["maps"]) > 0


def test_c2_baseline_size_bit_exact(dev):
    """BASELINE configs[1] (C2) at full size -- 50 Mbp / 100 contigs + 0.5 Gbases of ONT-like reads, k32 w100 --
    every minimizer, mapping, hit and PAF record against the oracle (not only properties)."""
    W = synth.workload("C2", 1.0)
    chroms, cbuf, coff, _, _ = synth.make_assembly(1, W["n_chrom"], W["contigs_per_chrom"], W["contig_len"])
    rbuf, roff, _ = synth.make_reads(2, chroms, W["read_bases"], W["read_len"], W["sub"], W["ins"], W["dele"], lognormal_sigma=0.4)
    k, w = W["k"], W["w"]
    ctg_len = np.diff(coff).astype(np.uint32)
    rlen = np.diff(roff).astype(np.uint32)
    with dev.batch(cbuf, coff) as cb, dev.sketch(cb, k, w) as csk, dev.index(csk, ctg_len) as ix, \
            dev.batch(rbuf, roff) as rb, dev.sketch(rb, k, w) as rsk, dev.map(ix, rsk, rlen, k=k, z=1000) as res:
        got = res.download()
        c_off, ch, cp, cs = csk.download()
        r_off, rh, rp, rs = rsk.download()
        nix = len(ix)
    o_off, oh, op, os_ = oracle.sketch_batch(cbuf, coff, k, w)
    assert np.array_equal(c_off, o_off) and np.array_equal(ch, oh) and np.array_equal(cp, op) and np.array_equal(cs, os_)
    q_off, qh, qp, qs = oracle.sketch_batch(rbuf, roff, k, w)
    assert np.array_equal(r_off, q_off) and np.array_equal(rh, qh) and np.array_equal(rp, qp) and np.array_equal(rs, qs)
    from helpers import contig_ids
    oix = oracle.Index(oh, contig_ids(o_off), op, os_)
    assert nix == len(oix)
    exp = oracle.map_reads(oix, ctg_len, q_off, rlen, qh, qp, qs, k=k, z=1000, threads=0)
    pc.assert_same_records(got, exp)
    assert len(got["maps"]) > 40000 and len(rh) > 9_000_000


def test_batch_beyond_2_pow_32_bases(dev):
    """One batch of 4.6 Gbases: global base indices exceed 2^32 (64-bit index arithmetic in every kernel).
    The reads beyond the 2^32 boundary, and a slice straddling it, are checked against the oracle."""
    rng = np.random.default_rng(11)
    unit = synth.random_bases(rng, 64_000_000)           # reads are windows of a 64 Mbp pool: cheap to generate
    n, rl = 230_000, 20_000
    starts = rng.integers(0, len(unit) - rl, n)
    buf = np.empty(n * rl, np.uint8)
    for i in range(n):
        buf[i * rl:(i + 1) * rl] = unit[starts[i]:starts[i] + rl]
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(rl))
    assert int(off[-1]) > 2 ** 32 + 200_000_000
    k, w = 32, 250
    with dev.batch(buf, off) as b, dev.sketch(b, k, w) as sk:
        moff, h, p, s = sk.download()
    first_beyond = int(np.searchsorted(off, 2 ** 32))
    for lo, hi in ((first_beyond - 3, first_beyond + 40), (n - 60, n), (0, 20)):
        sub = buf[int(off[lo]):int(off[hi])]
        soff = off[lo:hi + 1] - off[lo]
        ooff, oh, op, os_ = oracle.sketch_batch(sub, soff, k, w)
        a, bnd = int(moff[lo]), int(moff[hi])
        assert np.array_equal(moff[lo:hi + 1] - moff[lo], ooff)
        assert np.array_equal(h[a:bnd], oh) and np.array_equal(p[a:bnd], op) and np.array_equal(s[a:bnd], os_)
    dens = len(h) / float(off[-1])
    assert abs(dens - 2.0 / (w + 1)) < 0.0005


def test_anchor_function_matches_reference(dev):
    """ntlink_amd.anchor.get_accepted_anchor_contigs (gap-fill re-mapping entry point, bin/ntlink_utils.py:200-268)
    against vectors produced by the imported reference."""
    assert pc.check_anchor_cases(dev) == 125


# ---------------------------------------------------------------- round 2: the 32-bit window pass and its exact fallback

def _rand_seq(rng, n):
    return bytes(synth.random_bases(rng, n))


def test_fast_window_pass_decides_random_sequence_alone(dev, monkeypatch):
    """sketch_fast_kernel: on random sequence no strip needs the exact pass (and the sketch is the oracle's)."""
    monkeypatch.setenv("NTL_SKETCH_THRESH", "0")  # the threshold pass gives up strips with a candidate-free window: next test
    rng = np.random.default_rng(5)
    seqs = [_rand_seq(rng, n) for n in (90000, 4200, 170000, 300, 131, 5000, 1_000_000)]
    for k, w in ((32, 100), (32, 250), (24, 100), (40, 31), (20, 16), (20, 33), (64, 64), (100, 70), (32, 1000), (17, 3000)):
        with dev.batch(seqs) as b, dev.sketch(b, k, w) as sk:
            # 1.3 M windows x 7 / 2^32 chances of an entering key within SK2_NEAR of the minimum: none expected
            assert sk.strips > 0 and sk.redo_strips <= 1, (k, w, sk.strips, sk.redo_strips)
        pc.check_sketch(dev, seqs, k, w)


def test_wave_kernel_takes_the_large_windows(dev, monkeypatch):
    """Round 6: windows of 256 .. 1135 k-mers (large-genome runs of ntLink use w = 500, 1000) go through
    sketch_wave_kernel and per-strip lists like the windows up to 255 -- a window only enters that kernel's scans as a distance --, the
    strips it gives up through the two-level form of the block-minima pass: the oracle's sketch on random sequence, on the fuzz
    sequences (ties, N patterns, boundary lengths) and with every strip forced through all passes."""
    import fuzz_cases
    rng = np.random.default_rng(17)
    seqs = [_rand_seq(rng, n) for n in (90000, 4200, 700000, 1300, 131, 5000, 2_000_000)]
    for k, w in ((32, 256), (32, 500), (40, 1000), (24, 1135), (64, 300)):
        st = {}
        pc.check_sketch(dev, seqs, k, w, info=st)
        assert st["from_lists"] and st["strips"] > 0 and st["redo_strips"] <= 1 and st["fallback_strips"] < 0.05 * st["strips"] + 3, (k, w, st)
    for seed, k, w in ((1, 32, 500), (2, 21, 1000), (3, 24, 260)):
        pc.check_sketch(dev, fuzz_cases.fuzz_sequences(seed), k, w)
    monkeypatch.setenv("NTL_SKETCH_FORCE_REDO", "1")
    pc.check_sketch(dev, seqs[:3], 32, 500)
    monkeypatch.delenv("NTL_SKETCH_FORCE_REDO")
    monkeypatch.setenv("NTL_SKETCH_WAVE", "0")  # without the wave kernel such windows take the block-minima pass and the bitmask, as before
    st = {}
    pc.check_sketch(dev, seqs[:3], 32, 500, info=st)
    assert not st["from_lists"]


def test_threshold_window_pass_random_sequence(dev, monkeypatch):
    """sketch_thresh_kernel (the default for 71 <= w <= 255): the oracle's sketch on random sequence; about N p e^(-w p) of
    the strips (N p candidates per strip) have a window without a candidate and are decided by the block-minima pass
    (sketch_fast_list_kernel): 0.7 % at 10 candidates per window, most strips at 4, none with the pass switched off; the
    exact pass sees (next to) none of them."""
    rng = np.random.default_rng(7)
    seqs = [_rand_seq(rng, n) for n in (90000, 4200, 700000, 300, 131, 5000, 3_000_000)]
    frac = {}
    for cpw in ("10", "4", "13", "0"):
        monkeypatch.setenv("NTL_SKETCH_THRESH", cpw)
        for k, w in ((32, 250), (24, 121), (40, 255), (20, 180), (24, 100), (32, 71)):
            with dev.batch(seqs) as b, dev.sketch(b, k, w) as sk:
                frac[cpw, w] = sk.fallback_strips / sk.strips
                assert sk.redo_strips <= 1, (cpw, k, w, sk.redo_strips)
            pc.check_sketch(dev, seqs, k, w)
    assert frac["10", 250] < 0.03 and frac["4", 250] > 0.3 and frac["13", 250] < 0.005 and frac["0", 250] == 0, frac
    assert frac["10", 100] < 0.03 and frac["4", 100] > 0.3 and frac["0", 100] == 0, frac
    monkeypatch.setenv("NTL_SKETCH_THRESH", "10")
    monkeypatch.setenv("NTL_SKETCH_THRESH_DIRECT", "1")
    pc.check_sketch(dev, seqs, 32, 250)


def test_fast_window_pass_hands_ties_to_the_exact_pass(dev, monkeypatch):
    """Identical k-mers inside one window tie on the 32-bit key: detected, redone by the exact 64-bit pass."""
    rng = np.random.default_rng(6)
    unit = _rand_seq(rng, 37)
    seqs = [unit * 1500, b"A" * 30000, _rand_seq(rng, 25000) + unit * 40 + _rand_seq(rng, 25000), _rand_seq(rng, 60000),
            b"AC" * 12000 + _rand_seq(rng, 7000)]
    for k, w in ((32, 100), (24, 40), (32, 250), (7, 47)):
        with dev.batch(seqs) as b, dev.sketch(b, k, w) as sk:
            assert 0 < sk.redo_strips < sk.strips
        pc.check_sketch(dev, seqs, k, w)


def test_fast_window_pass_flags_keys_it_cannot_order(dev):
    """The window pass rolls only the hashes' 31-bit rings (bits 33..63): two different k-mers whose hashes agree in those
    bits have keys it cannot order.  Sequences in which such a pair competes for a window's minimum (one strip each) must all
    go to the exact pass, and the sketch is the oracle's whichever of the two is smaller."""
    for k, n in ((16, 48), (21, 24)):
        seqs = pc.near_tie_sequences(k, n, seed=11 + k)
        for w in (40, 64):
            st = {}
            pc.check_sketch(dev, seqs, k, w, info=st)
            assert st["redo_strips"] == st["strips"] == len(seqs), (k, w, st)
        seqs = pc.near_tie_sequences(k, n, seed=11 + k, third=True)  # a smaller k-mer right behind the pair: see parity_cases
        st = {}
        pc.check_sketch(dev, seqs, k, 40, info=st)
        assert st["redo_strips"] == st["strips"] == len(seqs), (k, st)


@pytest.mark.parametrize("env", [{"NTL_SKETCH_FORCE_REDO": "1"}, {"NTL_SKETCH_FAST": "0"}, {"NTL_SKETCH_NT": "128"}, {"NTL_SKETCH_NT": "256"},
                                 {"NTL_SKETCH_NT": "128", "NTL_SKETCH_FAST": "0"}, {"NTL_SKETCH_C": "4"}, {"NTL_SKETCH_CAP_GUESS": "1000"},
                                 {"NTL_SKETCH_LISTS": "0"}, {"NTL_LIST_SLOT": "8"}, {"NTL_LIST_SLOT": "8", "NTL_LIST_POOL": "2000"}],
                         ids=lambda e: ",".join(f"{k[11:]}={v}" for k, v in e.items()))
def test_sketch_variants_on_the_gpu(dev, monkeypatch, env):
    """Every tuning variant of the window pass and the second emit pass (record array denser than guessed) on the GPU,
    on the fuzz sequences (ties, N patterns, boundary lengths) and the edge cases."""
    import fuzz_cases
    for kk, vv in env.items():
        monkeypatch.setenv(kk, vv)
    for seed, k, w in ((1, 32, 100), (2, 32, 250), (3, 24, 100), (7, 40, 16)):
        pc.check_sketch(dev, fuzz_cases.fuzz_sequences(seed), k, w)
    pc.check_sketch(dev, pc.edge_sequences(), 32, 100)


def test_strip_lists(dev, monkeypatch):
    """Round 5: per-strip minimizer lists instead of the bitmask, every pass that writes them, lists in the pool, a pool that runs
    out (parity_cases.check_strip_lists), on sequences of up to 300 strips."""
    pc.check_strip_lists(dev, monkeypatch, scale=40)


@pytest.mark.parametrize("name,n_reads", [("C3", 100_000), ("C5", 100_000)])
def test_full_size_assembly_parity(dev, name, n_reads):
    """BASELINE configs[2] / configs[4] at FULL assembly size: 3 Gbp in 5000 contigs on the device (same generator as
    bench.py), every contig minimizer and the index size against the oracle, then 100 k sampled reads (ONT 15 kb at
    k32 w250; HiFi 20 kb at k24 w100 --sensitive) mapped on both sides, records byte-equal."""
    wl = synth.DeviceWorkload(dev, name, with_reads=False)
    W = wl.W
    k, w = W["k"], W["w"]
    cbuf, coff = wl.contigs.download()
    assert int(coff[-1]) > 2_900_000_000 and len(coff) - 1 == 5000
    with dev.sketch(wl.contigs, k, w) as csk, dev.index(csk, wl.ctg_len) as ix:
        # 3e9 windows x 7 / 2^32 near-ties of the ring keys + as many among the searched windows: about a dozen; the strips the
        # threshold pass gives up (a window without a candidate: 0.7 % of them) are decided by the block-minima pass
        assert csk.redo_strips < 64 and csk.fallback_strips < 0.015 * csk.strips, (csk.redo_strips, csk.fallback_strips, csk.strips)
        c_off, ch, cp, cs = csk.download()
        o_off, oh, op, os_ = oracle.sketch_batch(cbuf, coff, k, w)
        assert np.array_equal(c_off, o_off) and np.array_equal(ch, oh) and np.array_equal(cp, op) and np.array_equal(cs, os_)
        del ch, cp, cs, c_off, cbuf
        from helpers import contig_ids
        oix = oracle.Index(oh, contig_ids(o_off), op, os_)
        assert len(ix) == len(oix) > 20_000_000
        rb, rlen = wl.make_reads(n_reads * W["read_len"], seed=(5, 1))
        rbuf, roff = rb.download()
        params = dict(k=k, z=1000, x=0.0, sensitive=W["sensitive"], repeat_filter=False)
        with dev.sketch(rb, k, w) as rsk, dev.map(ix, rsk, rlen, **params) as res:
            got = res.download()
            r_off, rh, rp, rs = rsk.download()
        # the call form of the pair driver and of bench.py (round 6): a read sketch made only to be mapped
        # (ntl_sketch_run_for_map: no records, the lookups inside emit_list_kernel on the tagged full-size index), against the ORACLE
        with dev.sketch(rb, k, w, index=ix, records=False) as fsk, dev.map(ix, fsk, rlen, **params) as fres:
            assert not fsk.has_records
            got_for_map = fres.download()
        rb.close()
    q_off, qh, qp, qs = oracle.sketch_batch(rbuf, roff, k, w)
    assert np.array_equal(r_off, q_off) and np.array_equal(rh, qh) and np.array_equal(rp, qp) and np.array_equal(rs, qs)
    exp = oracle.map_reads(oix, wl.ctg_len, q_off, rlen, qh, qp, qs, threads=0, **params)
    pc.assert_same_records(got, exp)
    pc.assert_same_records(got_for_map, exp)
    assert len(got["maps"]) > 0.9 * len(rlen) and len(rlen) >= n_reads * 0.99
    wl.close()


def test_synth_generator_on_the_gpu(dev):
    """The device-side workload generator (bench / full-size tests): exact slices, reverse complements, determinism,
    error rates of the read model."""
    g = dev.synth_genome(1, [500_000, 300_000, 17])
    buf, off = g.download()
    cnt = np.bincount(buf, minlength=128)[[65, 67, 71, 84]]
    assert cnt.sum() == 800_017 and cnt.min() > 0.24 * cnt.sum()
    rng = np.random.default_rng(0)
    n = 300
    ln = rng.integers(1, 20000, n); sq = rng.integers(0, 2, n); rv = rng.integers(0, 2, n).astype(np.uint8)
    st = (rng.random(n) * (np.array([500_000, 300_000])[sq] - ln)).astype(np.int64)
    s = dev.synth_slices(g, 7, sq, st, ln, rv)
    sb, so = s.download()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for i in range(n):
        src = bytes(buf[int(off[sq[i]]) + st[i]: int(off[sq[i]]) + st[i] + ln[i]])
        assert bytes(sb[int(so[i]):int(so[i + 1])]) == (src.translate(comp)[::-1] if rv[i] else src)
    ln2 = np.full(50, 15000); st2 = rng.integers(0, 400_000, 50); sq2 = np.zeros(50, int); rv2 = rng.integers(0, 2, 50).astype(np.uint8)
    r1 = dev.synth_slices(g, 9, sq2, st2, ln2, rv2, sub=0.02, ins=0.015, dele=0.015)
    r2 = dev.synth_slices(g, 9, sq2, st2, ln2, rv2, sub=0.02, ins=0.015, dele=0.015)
    a, ao = r1.download()
    assert np.array_equal(a, r2.download()[0]) and int(ao[-1]) == 50 * 15000
    # about (1 - 0.05)^21 of the 21-mers of a read survive: its sketch must still hit the source
    with dev.sketch(g, 21, 10) as gsk, dev.index(gsk, [500_000, 300_000, 17]) as ix, dev.sketch(r1, 21, 10) as rsk, \
            dev.map(ix, rsk, ln2, k=21, z=10) as res:
        assert 0.2 < res.n_index_hits / rsk.count < 0.6
    for h in (g, s, r1, r2):
        h.close()


@pytest.mark.parametrize("case", pc.overlap_cases())
def test_overlap_consumer_matches_reference(dev, case, tmp_path):
    """the overlap stage's read_minimizers / read_minimizers_path (valid regions + per-contig duplicate
    removal, bin/ntlink_overlap_sequences.py:145-190) on the GPU == the imported reference on the same TSV."""
    assert pc.check_overlap_case(dev, case, tmp_path) > 500


@pytest.mark.parametrize("seed,nseq,max_len", [(1, 40, 30000), (2, 400, 5000), (3, 6, 2_000_000)])
def test_overlap_filter_random(dev, seed, nseq, max_len):
    kept, total = pc.check_overlap_random(dev, seed, nseq, max_len)
    assert kept < total


def test_probe_with_and_without_tags(dev):
    """probe_kernel<true> (first batch on an index) and probe_kernel<false> (after a batch that found most of its minimizers,
    the HiFi case): both equal the oracle, specific and sensitive."""
    rng = np.random.default_rng(12)
    contigs = [bytes(synth.random_bases(rng, n)) for n in (300_000, 120_000, 80_000, 999)]
    reads = [contigs[0][5000:45000], contigs[1][100:39000] + contigs[2][:20000], bytes(synth.random_bases(rng, 30000)),
             contigs[0][100_000:160_000], contigs[3] + contigs[2][40000:70000]] * 20
    for sens in (False, True):
        fr = pc.check_probe_forms(dev, contigs, reads, 24, 100, z=1000, sensitive=sens)
        assert min(fr) > 0.5


def test_async_order_and_unseen_results(dev, monkeypatch):
    """The queue-only calls on the real device, two streams: handles destroyed early, results asked for one batch late (so the
    window kernels of a batch really run beside the previous batch's lookup / map kernels); then with a record array that is
    too small, and the batch nobody asks about, whose overflow the next sync must report."""
    chroms, cbuf, coff, names, _ = synth.make_assembly(3, 1, 10, 200_000)
    rbuf, roff, _ = synth.make_reads(4, chroms, 5_000_000, 12_000, 0.02, 0.015, 0.015, lognormal_sigma=0.4)
    contigs = [cbuf[int(coff[i]):int(coff[i + 1])].tobytes() for i in range(len(coff) - 1)]
    reads = [rbuf[int(roff[i]):int(roff[i + 1])].tobytes() for i in range(len(roff) - 1)]
    assert dev.pipelined
    assert pc.check_async_order(dev, contigs, reads, 32, 100, z=1000) > 100
    monkeypatch.setenv("NTL_SKETCH_CAP_GUESS", "1000")
    with pytest.raises(capi.NtlError, match="destroyed before anybody asked"):
        pc.check_async_order(dev, contigs, reads, 32, 100, z=1000)
    dev.sync()


def test_preparation_on_a_stream_of_its_own(monkeypatch):
    """NTL_PREP_STREAM=1 (an experiment that is not the default): a sketch's per-sequence tables and strip table
    are made on a third stream while the previous window kernel still runs; their arrays go round in the cross-stream block cache
    (a block whose other streams are still busy is not waited for, another one is made).  Same records, same mappings."""
    monkeypatch.setenv("NTL_PREP_STREAM", "1")
    chroms, cbuf, coff, names, _ = synth.make_assembly(3, 1, 10, 200_000)
    rbuf, roff, _ = synth.make_reads(4, chroms, 5_000_000, 12_000, 0.02, 0.015, 0.015, lognormal_sigma=0.4)
    contigs = [cbuf[int(coff[i]):int(coff[i + 1])].tobytes() for i in range(len(coff) - 1)]
    reads = [rbuf[int(roff[i]):int(roff[i + 1])].tobytes() for i in range(len(roff) - 1)]
    d = capi.Device(0)
    try:
        assert d.pipelined
        assert pc.check_async_order(d, contigs, reads, 32, 250, z=1000) > 100
        pc.check_full_pipeline(d, contigs, reads[:60], 24, 100, z=1000, sensitive=True)
        d.sync()
    finally:
        d.close()


def test_handles_outlive_their_inputs(dev, monkeypatch):
    """ADVICE r3 on the device: the index, the contig sketch and both batches are destroyed before the result of a read sketch
    that overflowed its record array is asked for (sketch and mapping are then made again from that index), and 700 completed
    sketches + 700 completed map results stay alive at once on a context with 512 page-locked slots."""
    chroms, cbuf, coff, names, _ = synth.make_assembly(3, 1, 10, 200_000)
    rbuf, roff, _ = synth.make_reads(4, chroms, 3_000_000, 12_000, 0.02, 0.015, 0.015, lognormal_sigma=0.4)
    contigs = [cbuf[int(coff[i]):int(coff[i + 1])].tobytes() for i in range(len(coff) - 1)]
    reads = [rbuf[int(roff[i]):int(roff[i + 1])].tobytes() for i in range(len(roff) - 1)]
    monkeypatch.setenv("NTL_SKETCH_CAP_GUESS", "1000")
    assert pc.check_handles_outlive_their_inputs(dev, contigs, reads, 32, 100, z=1000, n_live=700) > 50


@pytest.mark.parametrize("k,w,rl,sens,err", [(32, 250, 15000, False, (0.02, 0.015, 0.015)), (24, 100, 20000, True, (0.001, 0.0005, 0.0005))],
                         ids=["C3_like", "C5_like"])
def test_text_made_on_the_device(dev, k, w, rl, sens, err):
    """ntl_mapres_format on scaled-down C3- and C5-like workloads (thousands of mappings, 10^5 .. 10^6 hits, names of different
    lengths): the device's bytes == the host emitters' == the oracle's; the pair tally's hit ends == the records'."""
    chroms, cbuf, coff, names, _ = synth.make_assembly(5, 2, 8, 250_000)
    contigs = [cbuf[int(coff[i]):int(coff[i + 1])].tobytes() for i in range(len(coff) - 1)]
    rbuf, roff, _ = synth.make_reads(6, chroms, 20_000_000, rl, *err, lognormal_sigma=0.4)
    reads = [rbuf[int(roff[i]):int(roff[i + 1])].tobytes() for i in range(len(roff) - 1)]
    n, nv, npf = pc.check_device_text(dev, contigs, reads, k, w, z=1000, sensitive=sens)
    assert n > 500 and nv > 100_000 and npf > 10_000
    assert pc.check_device_text(dev, contigs[:2], [b"ACGT" * 500, b""], k, w, z=1000) == (0, 0, 0)


def test_one_stream_and_back(dev):
    contigs = pc.fixture_seqs("scaffolds_1.fa")
    reads = pc.fixture_seqs("long_reads_4_top5.fa")
    dev.set_pipeline(False)
    assert not dev.pipelined
    pc.check_full_pipeline(dev, contigs, reads, 32, 250, z=1000)
    dev.set_pipeline(True)
    assert dev.pipelined
    pc.check_full_pipeline(dev, contigs, reads, 32, 250, z=1000, sensitive=True)


@pytest.mark.parametrize("env", [{"NTL_SKETCH_THRESH": "0"}, {"NTL_SKETCH_THRESH": "0", "NTL_SKETCH_LANES": "1"}, {"NTL_EMIT_U": "2"},
                                 {"NTL_SKETCH_THRESH": "0", "NTL_SKETCH_LANES": "1", "NTL_EMIT_U": "2"}, {"NTL_SKETCH_THRESH": "5"}, {"NTL_SKETCH_THRESH": "13"},
                                 {"NTL_SKETCH_WAVE": "0"}, {"NTL_SKETCH_WAVE": "2"}, {"NTL_SKETCH_WAVE": "8"}, {"NTL_SKETCH_WAVE": "9"},
                                 {"NTL_SKETCH_WAVE": "0", "NTL_SKETCH_THRESH": "5"}],
                         ids=lambda e: ",".join(f"{k[4:]}={v}" for k, v in e.items()))
def test_kernel_variants_full_pipeline(dev, monkeypatch, env):
    """The window passes that are not the default for 71 <= w <= 255 (sketch_fast_kernel; sketch_lanes_kernel, the experiment of
    DESIGN 4.12), the threshold pass with other candidate densities than the default, and the two-wide emit kernel on the GPU:
    scaled-down C3- and C5-like workloads, full pipeline against the oracle, and the fuzz sequences."""
    import fuzz_cases
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    chroms, cbuf, coff, names, _ = synth.make_assembly(1, 1, 12, 250_000, n_run_every=5)
    contigs = [cbuf[int(coff[i]):int(coff[i + 1])].tobytes() for i in range(len(coff) - 1)]
    for k, w, rl, sens, err in ((32, 250, 15000, False, (0.02, 0.015, 0.015)), (24, 100, 20000, True, (0.001, 0.0005, 0.0005))):
        rbuf, roff, _ = synth.make_reads(2, chroms, 6_000_000, rl, *err, lognormal_sigma=0.4)
        reads = [rbuf[int(roff[i]):int(roff[i + 1])].tobytes() for i in range(len(roff) - 1)]
        pc.check_full_pipeline(dev, contigs, reads, k, w, z=1000, sensitive=sens)
    for seed, k, w in ((1, 32, 100), (2, 32, 250), (3, 24, 64), (5, 40, 255)):
        pc.check_sketch(dev, fuzz_cases.fuzz_sequences(seed), k, w)


def test_slab_blocks_are_reused_when_the_cache_drops_them(monkeypatch):
    """blocks of 192 MB or less are cut from 1-GB slabs, and a slab block that the block cache evicted used to be lost
    (the cache's bound no longer bounded device memory).  With a cache bound of zero every released block is evicted at once: the
    slabs' free list hands them out again, so sixty sketches of varying size make no more allocations than the first few did --
    and stay bit-exact."""
    monkeypatch.setenv("NTL_POOL_MAX_BYTES", "0")
    d = capi.Device(0)
    try:
        rng = np.random.default_rng(3)
        acgt = np.frombuffer(b"ACGT", np.uint8)
        counts = []
        for i in range(60):
            n = int(rng.integers(20_000, 400_000))
            seqs = [bytes(acgt[rng.integers(0, 4, n)]), bytes(acgt[rng.integers(0, 4, n // 3)])]
            assert pc.check_sketch(d, seqs, 32, 250 if i % 2 else 100) > 0
            counts.append(d.prof_get("hipMalloc")[1])
        assert counts[-1] <= counts[9] + 1, counts  # (single blocks above 192 MB would count too: none here)
    finally:
        d.close()


@pytest.mark.parametrize("name", ["C3", "C5"])
def test_full_sub_batch_lists_equal_bitmask(name, monkeypatch):
    """A size-independent check at the bench's own sub-batch size (3.9 Gbases of device-generated reads against the full 3-Gbp
    assembly): the window -> emit stage exists twice -- per-strip lists + emit_list_kernel, and the bitmask + mask_count / emit_kernel
    (NTL_SKETCH_LISTS=0) -- with different ownership rules, ranks and kernels; every minimizer record of the contigs and of the
    reads, the index size and every mapping / hit / PAF record of the batch must agree between the two."""
    d = capi.Device(0)
    try:
        wl = synth.DeviceWorkload(d, name, with_reads=False)
        k, w = wl.W["k"], wl.W["w"]
        rb, rlen = wl.make_reads(3_900_000_000, seed=(77, 1))
        got = []
        for lists in ("1", "0"):
            monkeypatch.setenv("NTL_SKETCH_LISTS", lists)
            with d.sketch(wl.contigs, k, w) as csk, d.index(csk, wl.ctg_len) as ix, d.sketch(rb, k, w, index=ix) as rsk, \
                    d.map(ix, rsk, rlen, k=k, z=1000, x=0.0, sensitive=wl.W["sensitive"], repeat_filter=False) as res:
                assert csk.from_lists == rsk.from_lists == (lists == "1")
                r = res.download()
                got.append((csk.download(), len(ix), rsk.download(), r["maps"], r["hits"], r["pafs"]))
        (c1, n1, s1, m1, h1, p1), (c0, n0, s0, m0, h0, p0) = got
        assert n1 == n0 and len(s1[1]) == len(s0[1]) > 20_000_000
        for a, b in zip(c1 + s1, c0 + s0):
            assert np.array_equal(a, b)
        assert np.array_equal(m1, m0) and np.array_equal(h1, h0) and np.array_equal(p1, p0) and len(m1) > 100_000
        rb.close(); wl.close()
    finally:
        d.close()

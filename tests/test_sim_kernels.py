"""The product's kernel source executed under the SIMT mock (tests/sim: one pthread per lane), checked
against the oracle.  This is a CPU debugging aid for the kernels' logic and for sanitizers -- the parity
tests proper are tests/test_gpu_parity.py on a real MI355X."""
import numpy as np
import pytest

import parity_cases as pc
from sim import simlib


@pytest.fixture(scope="module")
def dev():
    d = simlib.device()
    yield d
    d.close()


@pytest.mark.parametrize("fa,k,w", [("scaffolds_2.fa", 32, 100), ("scaffolds_4.fa", 40, 100), ("scaffolds_1.fa", 32, 250)])
def test_sim_sketch_fixtures(dev, fa, k, w):
    assert pc.check_sketch(dev, pc.fixture_seqs(fa), k, w) > 0


@pytest.mark.parametrize("k,w", [(32, 100), (8, 4), (5, 17), (3, 1), (32, 250)])
def test_sim_sketch_edges(dev, k, w):
    pc.check_sketch(dev, pc.edge_sequences(), k, w)


def test_sim_sketch_second_emit_pass_when_denser_than_guessed(dev, monkeypatch):
    """The record array is sized from the expected density before the count is known; a denser batch is
    emitted again into an exact-size array."""
    monkeypatch.setenv("NTL_SKETCH_CAP_GUESS", "7")
    pc.check_sketch(dev, pc.edge_sequences(), 32, 100)
    pc.check_sketch(dev, [b"ACGTTGCA" * 400, b"AC" * 900], 8, 40)


def test_sim_sketch_many_tiny_sequences(dev):
    pc.check_sketch(dev, pc.tiny_sequences(700), 12, 8)


@pytest.mark.parametrize("w", [2, 5, 10, 15])
def test_sim_small_window_pass(dev, w):
    """sketch_small_kernel<W> (2 <= w <= 15, round 6) under the mock: strips of 4096 elements with 4079 own windows, ties, N runs."""
    seqs = pc.small_window_sequences(n_long=1, long_len=9000)
    assert pc.check_small_windows(dev, (w,), ks=(15, 33) if w == 5 else (20,), seqs=seqs, fuzz_seeds=(4,)) > 0


def test_sim_sketch_reads_small_w(dev):
    reads = pc.fixture_seqs("long_reads_4_top5.fa")
    pc.check_sketch(dev, reads[:2], 15, 5)
    pc.check_sketch(dev, reads, 40, 100)


@pytest.mark.parametrize("name", ["syn_default", "syn_sensitive", "syn_sens_repeat", "syn_x03", "syn_many_ctg"])
def test_sim_map_scenarios(dev, name):
    pc.check_scenario(dev, name)


def test_sim_anchor_function_matches_reference(dev):
    """get_accepted_anchor_contigs with the reference's signature, on the device."""
    assert pc.check_anchor_cases(dev, max_cases=12) >= 10


def test_sim_full_pipeline_top5(dev):
    got = pc.check_full_pipeline(dev, pc.fixture_seqs("scaffolds_4.fa"), pc.fixture_seqs("long_reads_4_top5.fa"),
                                 40, 100, z=1000)
    assert len(got["pafs"]) == 6


def test_sim_pair_driver_files(dev, tmp_path):
    """The fused `ntLink pair` driver over the SIMT mock leaves the reference's files (top-5 reads)."""
    import os
    import shutil
    from helpers import GEN, REF, TEST7_PAF, read_text
    from ntlink_amd import pipeline
    for n in ("scaffolds_4.fa", "long_reads_4_top5.fa"):
        shutil.copy(os.path.join(REF, n), tmp_path / n)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        pipeline.run_pair(dev, "scaffolds_4.fa", "long_reads_4_top5.fa", k=40, w=100, paf=True, pairs_tsv=True)
        # one read per batch, three device worker threads (contexts) taking batches in turn: results still in read order
        os.environ["NTL_DEVICE_STREAMS"] = "3"
        try:
            st = pipeline.run_pair(dev, "scaffolds_4.fa", "long_reads_4_top5.fa", k=40, w=100, paf=True, pairs_tsv=True,
                                   batch_bases=1000, prefix="streams3", write_contig_tsv=False)
        finally:
            del os.environ["NTL_DEVICE_STREAMS"]
        assert st["reads"] == 5
        # a batch whose text may pass the 32-bit offsets of the device's formatter (here: pretended, 1000 bytes) is refused with
        # NTL_ERANGE -- headers, tokens and PAF lines each bounded -- and the host's emitters write the same bytes
        os.environ["NTL_FORMAT_MAX_TEXT"] = "1000"
        try:
            from ntlink_amd import capi
            ctgs, rds = pc.fixture_seqs("scaffolds_4.fa"), pc.fixture_seqs("long_reads_4_top5.fa")
            cl, rl = np.array([len(x) for x in ctgs], np.uint32), np.array([len(x) for x in rds], np.uint32)
            with dev.batch(ctgs) as cb, dev.sketch(cb, 40, 100) as csk, dev.index(csk, cl) as ix, dev.batch(rds) as rb, \
                    dev.sketch(rb, 40, 100, index=ix) as rsk, dev.map(ix, rsk, rl, k=40, z=1000) as res, \
                    dev.names(["r%d" % i for i in range(len(rds))], rl) as rn, dev.names(["c%d" % i for i in range(len(ctgs))], cl) as cn:
                with pytest.raises(capi.NtlError) as ei:
                    res.format(rn, cn, True, True)
                assert ei.value.code == capi.NTL_ERANGE
            pipeline.run_pair(dev, "scaffolds_4.fa", "long_reads_4_top5.fa", k=40, w=100, paf=True, pairs_tsv=True, prefix="hostfmt", write_contig_tsv=False)
        finally:
            del os.environ["NTL_FORMAT_MAX_TEXT"]
    finally:
        os.chdir(cwd)
    pre = str(tmp_path / "scaffolds_4.fa.k40.w100.z1000")
    d = os.path.join(GEN, "fixtures", "t7_top5_k40_w100")
    for ext in (".verbose_mapping.tsv", ".paf", ".pairs.tsv"):
        assert read_text(str(tmp_path / "streams3") + ext) == read_text(d + ext), ext
        assert read_text(str(tmp_path / "hostfmt") + ext) == read_text(d + ext), ext
    assert read_text(pre + ".verbose_mapping.tsv") == read_text(d + ".verbose_mapping.tsv")
    assert set(read_text(pre + ".paf").splitlines()) == TEST7_PAF
    assert read_text(pre + ".pairs.tsv") == read_text(d + ".pairs.tsv")
    assert read_text(str(tmp_path / "scaffolds_4.fa.k40.w100.tsv")) == read_text(os.path.join(REF, "expected_outputs", "scaffolds_4.fa.k40.w100.tsv"))
    assert os.path.exists(pre + ".n1.scaffold.dot")


def test_sim_pair_driver_several_gzip_read_files(dev, tmp_path, monkeypatch):
    """`reads='a.fa.gz b.fa.gz c.fa.gz'` (ntLink:222: `gzip -cd -f FILES`) with two parallel readers: the run of gzip files is one
    chunk of theirs (inflated ahead, a thread per file) -- same files as from one plain FASTA."""
    import gzip
    import os
    import shutil
    from helpers import REF, read_text
    from ntlink_amd import pipeline
    for n in ("scaffolds_4.fa", "long_reads_4_top5.fa"):
        shutil.copy(os.path.join(REF, n), tmp_path / n)
    recs = open(tmp_path / "long_reads_4_top5.fa").read().split(">")[1:]
    assert len(recs) == 5
    for i, part in enumerate((recs[:2], recs[2:3], recs[3:])):
        with gzip.open(tmp_path / f"part{i}.fa.gz", "wt") as fh:
            fh.write("".join(">" + r for r in part))
    monkeypatch.setenv("NTL_IO_READERS", "2")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        pipeline.run_pair(dev, "scaffolds_4.fa", "long_reads_4_top5.fa", k=40, w=100, paf=True, pairs_tsv=True, prefix="plain", write_contig_tsv=False)
        st = pipeline.run_pair(dev, "scaffolds_4.fa", "part0.fa.gz part1.fa.gz part2.fa.gz", k=40, w=100, paf=True, pairs_tsv=True,
                               prefix="gz", batch_bases=20_000, write_contig_tsv=False)
        assert st["reads"] == 5
    finally:
        os.chdir(cwd)
    for ext in (".verbose_mapping.tsv", ".paf", ".pairs.tsv"):
        assert read_text(str(tmp_path / "gz") + ext) == read_text(str(tmp_path / "plain") + ext), ext


def test_sim_pair_driver_error_removes_partial_outputs(dev, tmp_path):
    """bin/ntlink_pair.py:608-613: on an error nothing half-written stays behind; the reader thread's
    exception surfaces in the driver."""
    import os
    import shutil
    from helpers import REF
    from ntlink_amd import pipeline
    shutil.copy(os.path.join(REF, "scaffolds_4.fa"), tmp_path / "scaffolds_4.fa")
    (tmp_path / "broken.fa.gz").write_bytes(b"\x1f\x8b\x08\x00garbage-not-a-gzip-stream" * 10)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        with pytest.raises(OSError):
            pipeline.run_pair(dev, "scaffolds_4.fa", "long_reads_missing.fa", k=40, w=100, paf=True)
        with pytest.raises(OSError):
            pipeline.run_pair(dev, "scaffolds_4.fa", "broken.fa.gz", k=40, w=100, paf=True)
    finally:
        os.chdir(cwd)
    left = sorted(os.listdir(tmp_path))
    assert not [f for f in left if f.endswith(".verbose_mapping.tsv") or f.endswith(".paf")], left


def test_sim_pinned_pool_reuses_buffers(dev):
    """Device.pinned_empty / pinned_release: best-fit reuse, no small request on a large buffer, release
    of foreign arrays is a no-op; the bytes are usable as ntl_batch_create input."""
    import numpy as np
    a = dev.pinned_empty(3_000_000)
    addr_a = a.ctypes.data
    a[:200_000] = np.frombuffer(b"ACGT", np.uint8)[np.arange(200_000) % 4]
    with dev.batch(a[:200_000], np.array([0, 50_000, 200_000], np.uint64)) as b:
        assert b.nseq == 2 and b.bases == 200_000
    dev.pinned_release(a)
    dev.pinned_release(np.zeros(4, np.uint8))
    b2 = dev.pinned_empty(2_900_000)
    assert b2.ctypes.data == addr_a  # reused
    c = dev.pinned_empty(1000)
    assert c.ctypes.data != addr_a
    dev.pinned_release(b2)
    d = dev.pinned_empty(1000)       # a 1 kB request must not take the 3 MB buffer
    assert d.ctypes.data != addr_a
    for x in (c, d):
        dev.pinned_release(x)


def test_sim_sketch_arrays_round_trip_threaded(dev):
    """ntl_sketch_from_host / ntl_sketch_download split their column <-> record conversion over
    threads above 2^18 minimizers."""
    import numpy as np
    rng = np.random.default_rng(9)
    nseq, n = 5000, 400_000
    cuts = np.sort(rng.integers(0, n + 1, nseq - 1))
    off = np.concatenate([[0], cuts, [n]]).astype(np.uint64)
    h = rng.integers(0, 2**63, n, dtype=np.uint64) * 2 + 1
    p = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    st = rng.integers(0, 2, n).astype(np.uint8)
    with dev.sketch_from_arrays(off, h, p, st) as sk:
        o2, h2, p2, s2 = sk.download()
    assert np.array_equal(o2, off) and np.array_equal(h2, h) and np.array_equal(p2, p) and np.array_equal(s2, st)


def test_sim_two_operator_files(dev, tmp_path):
    """Operator A then operator B2 through text, as the reference's Makefile joins them
    (ntLink:198-199,221-225): run_indexlr -> TSV files -> run_ntlink_pair (native TSV parser)."""
    import argparse
    import os
    from helpers import GEN, REF, TEST7_PAF, read_text
    from ntlink_amd import pipeline
    ctsv, rtsv = str(tmp_path / "c.tsv"), str(tmp_path / "r.tsv")
    with open(ctsv, "w") as fh:
        pipeline.run_indexlr(dev, [os.path.join(REF, "scaffolds_4.fa")], 40, 100, fh, False)
    with open(rtsv, "w") as fh:
        pipeline.run_indexlr(dev, [os.path.join(REF, "long_reads_4_top5.fa")], 40, 100, fh, True)
    assert read_text(ctsv) == read_text(os.path.join(REF, "expected_outputs", "scaffolds_4.fa.k40.w100.tsv"))
    pre = str(tmp_path / "out")
    pipeline.run_ntlink_pair(dev, argparse.Namespace(FILES=[rtsv], s=os.path.join(REF, "scaffolds_4.fa"), m=ctsv, p=pre, n=1, k=40,
                                                     z=1000, a=1, f=10, x=0.0, checkpoint=None, pairs=True, paf=True,
                                                     sensitive=False, repeat_filter=False, verbose=True))
    d = os.path.join(GEN, "fixtures", "t7_top5_k40_w100")
    assert read_text(pre + ".verbose_mapping.tsv") == read_text(d + ".verbose_mapping.tsv")
    assert set(read_text(pre + ".paf").splitlines()) == TEST7_PAF
    assert read_text(pre + ".pairs.tsv") == read_text(d + ".pairs.tsv")


@pytest.mark.parametrize("seed,k,w", [(21, 32, 100), (22, 80, 20), (23, 9, 3)])
def test_sim_fuzz_sketch(dev, seed, k, w):
    import fuzz_cases
    pc.check_sketch(dev, fuzz_cases.fuzz_sequences(seed, n=16, max_len=5000), k, w)


@pytest.mark.parametrize("seed", [0, 3])
def test_sim_fuzz_mapping(dev, seed):
    import fuzz_cases
    arrs = fuzz_cases.fuzz_mapping(seed, n_reads=60)
    kw = dict(k=24, z=[1000, 500, 1000, 1][seed % 4], x=[0.0, 0.0, 1.2, 0.4][seed % 4], sensitive=bool(seed & 1),
              repeat_filter=bool(seed & 2))
    pc.check_pair_arrays(dev, *arrs, **kw)


def _rand_seq(rng, n):
    return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))


def test_sim_fast_window_pass_decides_random_sequence_alone(dev, monkeypatch):
    """sketch_fast_kernel (32-bit keys, searched change points): on random sequence no strip needs the exact pass,
    and the result is the oracle's.  (The threshold pass, which gives a strip up when one of its windows has no candidate,
    is switched off: test_sim_threshold_window_pass.)"""
    monkeypatch.setenv("NTL_SKETCH_THRESH", "0")
    rng = np.random.default_rng(5)
    seqs = [_rand_seq(rng, n) for n in (9000, 4200, 300, 131, 5000)]
    for k, w in ((32, 100), (32, 250), (24, 100), (40, 31), (15, 16), (100, 70)):
        st = {}
        pc.check_sketch(dev, seqs, k, w, info=st)
        assert st["strips"] > 0 and st["redo_strips"] == 0, (k, w, st)
    st = {}
    pc.check_sketch(dev, seqs, 7, 47, info=st)  # 4^7 distinct k-mers: identical k-mers do share windows
    assert 0 < st["redo_strips"] < st["strips"]


def test_sim_wave_kernel_takes_the_large_windows(dev):
    """Windows of 256 .. 1135 k-mers through sketch_wave_kernel and the lists (round 6), under the mock."""
    rng = np.random.default_rng(18)
    seqs = [_rand_seq(rng, n) for n in (9000, 4200, 300, 1300, 14000)]
    for k, w in ((32, 500), (24, 1000), (32, 256)):
        st = {}
        pc.check_sketch(dev, seqs, k, w, info=st)
        assert st["from_lists"] and st["strips"] > 0, (k, w, st)


def test_sim_fast_window_pass_hands_ties_to_the_exact_pass(dev, monkeypatch):
    """Identical k-mers inside one window (tandem repeats, homopolymers) tie on the 32-bit key: those strips must be
    detected and redone by the exact 64-bit pass; forcing every strip through both passes changes nothing."""
    rng = np.random.default_rng(6)
    unit = _rand_seq(rng, 37)
    seqs = [unit * 150, b"A" * 3000, _rand_seq(rng, 2500) + unit * 40 + _rand_seq(rng, 2500), _rand_seq(rng, 6000),
            b"AC" * 1200 + _rand_seq(rng, 700)]
    st = {}
    for k, w in ((32, 100), (24, 40), (32, 250)):
        pc.check_sketch(dev, seqs, k, w, info=st)
        assert 0 < st["redo_strips"] < st["strips"]
    monkeypatch.setenv("NTL_SKETCH_FORCE_REDO", "1")
    pc.check_sketch(dev, seqs, 32, 100, info=st)
    assert st["redo_strips"] == st["strips"]
    monkeypatch.delenv("NTL_SKETCH_FORCE_REDO")
    monkeypatch.setenv("NTL_SKETCH_FAST", "0")
    pc.check_sketch(dev, seqs, 32, 100, info=st)
    assert st["redo_strips"] == 0


def test_sim_fast_window_pass_flags_keys_it_cannot_order(dev):
    """The window pass rolls only the hashes' 31-bit rings: two different k-mers whose hashes agree in bits 33..63 have keys
    it cannot order.  When such a pair competes for a window's minimum the strip must go to the exact pass (every sequence
    here is one strip), and the result is the oracle's whichever of the two comes first."""
    k = 16
    seqs = pc.near_tie_sequences(k, 24)
    for w in (40, 64):
        st = {}
        pc.check_sketch(dev, seqs, k, w, info=st)
        assert st["redo_strips"] == st["strips"] == len(seqs), st
    # ... and nothing the 32-bit pass decided for such a strip is kept: with a smaller k-mer right behind the pair, the later
    # one of the pair is a minimizer only if it is the smaller one
    seqs = pc.near_tie_sequences(k, 24, third=True)
    st = {}
    pc.check_sketch(dev, seqs, k, 40, info=st)
    assert st["redo_strips"] == st["strips"] == len(seqs), st


@pytest.mark.parametrize("case", ["synthetic_k15_w5_s1", "synthetic_k8_w3_s3", "scaffolds_4_k15_w5_s1"])
def test_sim_overlap_consumer_matches_reference(dev, case, tmp_path):
    """read_minimizers / read_minimizers_path of the overlap stage (valid regions, per-contig duplicate
    removal) on the device."""
    assert pc.check_overlap_case(dev, case, tmp_path) > 500


def test_sim_overlap_filter_random(dev):
    kept, total = pc.check_overlap_random(dev, 3, nseq=12, max_len=6000)
    assert kept < total


def test_sim_map_overflow_reads_take_the_global_scratch_kernel(dev):
    """More hits than the LDS staging holds, and few hits on more contigs than it has runs: map_overflow_kernel."""
    from ntlink_amd import synth
    rng = np.random.default_rng(3)
    contigs = [bytes(synth.random_bases(rng, 1500)) for _ in range(150)]
    order = rng.permutation(150)
    dense = b"".join(contigs[i] for i in order[:40])                  # > 1024 hits (the largest size class)
    patchy = b"".join(contigs[i][700:748] for i in order[:140])       # < 1024 hits, > 128 runs
    mid = b"".join(contigs[i][:1200] for i in order[40:46])          # 256 < minimizers <= 1024: the two larger LDS classes
    got = pc.check_full_pipeline(dev, contigs, [dense, patchy, contigs[3][:400], mid, mid[:4000]], 24, 20, z=1000)
    maps = got["maps"]
    assert int(maps["n_hits"][maps["read"] == 0].sum()) > 1024
    assert int((maps["read"] == 1).sum()) > 128 and int(maps["n_hits"][maps["read"] == 1].sum()) <= 1024
    assert 512 < int(maps["n_hits"][maps["read"] == 3].sum()) <= 1024 and 256 < int(maps["n_hits"][maps["read"] == 4].sum()) <= 512


def test_sim_packed_batches_sketch_like_ascii_batches(dev, tmp_path):
    """ntl_batch_create_packed (bases packed and ACGT runs found by the parser threads) == ntl_batch_create on the same
    records: sketches of both equal the oracle's (N runs, lower case, IUPAC, empty and tiny records)."""
    import oracle
    from ntlink_amd import seqio
    seqs = pc.edge_sequences() + [b"ACGTNNNNNACGTTGCAATGC" * 40, b"acgtacgtac" * 90 + b"R" + b"GATTACA" * 70]
    p = tmp_path / "e.fa"
    with open(p, "wb") as fh:
        for i, s in enumerate(seqs):
            fh.write(b">q%d\n" % i + s + b"\n")
    ss = seqio.load_all([str(p)], packed=True)
    assert len(ss) == len(seqs)
    for k, w in ((32, 100), (12, 8)):
        with dev.batch_packed(ss) as b, dev.sketch(b, k, w) as sk:
            off, h, q, s = sk.download()
        ooff, oh, op, os_ = oracle.sketch_batch(b"".join(seqs), pc.offsets_of(seqs), k, w)
        assert np.array_equal(off, ooff) and np.array_equal(h, oh) and np.array_equal(q, op) and np.array_equal(s, os_)


def test_sim_probe_with_and_without_tags(dev):
    """probe_kernel<true> (first batch) and probe_kernel<false> (after a batch that found most minimizers): same records."""
    rng = np.random.default_rng(12)
    contigs = [_rand_seq(rng, n) for n in (9000, 4000, 2500)]
    reads = [contigs[0][500:4000], contigs[1][100:3900] + contigs[2][:2000], _rand_seq(rng, 3000), contigs[0][3000:8000]]
    fr = pc.check_probe_forms(dev, contigs, reads, 24, 30, z=1000)
    assert min(fr) > 0.5


def test_sim_async_order_and_unseen_results(dev, monkeypatch):
    """Queue-only calls, handles destroyed early, results asked for one batch late; then the same with a record array that is
    too small (both the sketch and the map are made again when the count is finally asked for), and the case nobody asks:
    the overflow is reported by the next sync instead of vanishing."""
    from ntlink_amd import capi
    contigs = pc.fixture_seqs("scaffolds_4.fa")
    reads = pc.fixture_seqs("long_reads_4_top5.fa")
    assert pc.check_async_order(dev, contigs, reads, 40, 100, z=1000) > 0
    monkeypatch.setenv("NTL_SKETCH_CAP_GUESS", "40")
    with pytest.raises(capi.NtlError, match="destroyed before anybody asked"):
        pc.check_async_order(dev, contigs, reads, 40, 100, z=1000)  # the records compare equal (redone), the unseen batch is reported
    dev.sync()  # reported once


def test_sim_handles_outlive_their_inputs(monkeypatch):
    """the index destroyed before an overflowed sketch is made again from it; more live completed handles than the
    context has page-locked slots (a context of its own with four of them: the mock takes seconds per sketch)."""
    contigs = pc.fixture_seqs("scaffolds_4.fa")[:6]
    reads = pc.fixture_seqs("long_reads_4_top5.fa")[:2]
    monkeypatch.setenv("NTL_SKETCH_CAP_GUESS", "40")
    monkeypatch.setenv("NTL_NSLOTS", "4")
    d = simlib.device()
    try:
        pc.check_handles_outlive_their_inputs(d, contigs, reads, 40, 100, z=1000, n_live=6, tiny_len=700)
    finally:
        d.close()


def test_sim_preparation_on_a_stream_of_its_own(monkeypatch):
    """NTL_PREP_STREAM=1 under the mock: the third stream's code path (three block caches, blocks used on several streams going
    round) -- the mock runs streams in order, so this checks the bookkeeping, the GPU test the concurrency."""
    contigs = pc.fixture_seqs("scaffolds_4.fa")
    reads = pc.fixture_seqs("long_reads_4_top5.fa")
    monkeypatch.setenv("NTL_PREP_STREAM", "1")
    d = simlib.device()
    try:
        assert pc.check_async_order(d, contigs, reads, 40, 100, z=1000) > 0
        d.sync()
    finally:
        d.close()


def test_sim_text_made_on_the_device(dev):
    """ntl_mapres_format under the mock: the verbose and PAF lines of test 7's five reads == the host emitters' == the oracle's;
    and a batch without any mapping gives two empty texts."""
    contigs = pc.fixture_seqs("scaffolds_4.fa")
    reads = pc.fixture_seqs("long_reads_4_top5.fa")
    n, nv, npf = pc.check_device_text(dev, contigs, reads, 40, 100, z=1000)
    assert n >= 6 and nv > 100 and npf > 100
    assert pc.check_device_text(dev, contigs[:3], [b"ACGT" * 200, b"", b"N" * 300], 40, 100, z=1000) == (0, 0, 0)


def test_sim_one_stream_and_back(dev):
    """ntl_ctx_set_pipeline: the window stage back on the one stream at a quiet point, and out again: same records."""
    contigs = pc.fixture_seqs("scaffolds_4.fa")
    reads = pc.fixture_seqs("long_reads_4_top5.fa")
    assert dev.pipelined
    dev.set_pipeline(False)
    assert not dev.pipelined
    pc.check_full_pipeline(dev, contigs, reads, 40, 100, z=1000)
    dev.set_pipeline(True)
    assert dev.pipelined
    pc.check_full_pipeline(dev, contigs, reads, 40, 100, z=1000, sensitive=True)


@pytest.mark.parametrize("env", [{"NTL_SKETCH_THRESH": "0"}, {"NTL_SKETCH_THRESH": "0", "NTL_SKETCH_LANES": "1"}, {"NTL_EMIT_U": "2"},
                                 {"NTL_SKETCH_THRESH": "4"}, {"NTL_SKETCH_THRESH": "13"}])
def test_sim_kernel_variants_full_pipeline(dev, monkeypatch, env):
    """The window passes that are not the default for 71 <= w <= 255 (sketch_fast_kernel, sketch_lanes_kernel), the threshold
    pass with few candidates per window (most strips have a window without one and take the exact pass) and with many, and the
    emit kernel with two minimizers in flight per thread: same records as the oracle on fixtures, fuzz sequences (ties, N
    patterns) and windows of both 20-KB ranges."""
    import fuzz_cases
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    pc.check_full_pipeline(dev, pc.fixture_seqs("scaffolds_4.fa"), pc.fixture_seqs("long_reads_4_top5.fa"), 40, 100, z=1000)
    for seed, k, w in ((1, 32, 100), (2, 32, 250), (3, 24, 64), (4, 40, 130)):
        pc.check_sketch(dev, fuzz_cases.fuzz_sequences(seed)[:12], k, w)


def test_sim_threshold_window_pass(dev, monkeypatch):
    """sketch_thresh_kernel (the default window pass for 71 <= w <= 255; without staged keys below 121): random sequences of
    lengths around the strip and window sizes at the ends of its ranges of w, against the oracle.  With 4 candidates per window
    instead of 10 most strips have a window without a candidate: they come back from the block-minima pass
    (sketch_fast_list_kernel) with the same sketch, and none of them needs the exact pass; with NTL_SKETCH_THRESH=0 nothing
    takes the threshold pass."""
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seqs = [bytes(acgt[rng.integers(0, 4, n)]) for n in (60000, 5000, 300, 4127, 281, 282, 3871, 8000, 0, 31)]
    fb = {}
    for cpw in ("10", "4", "0"):
        monkeypatch.setenv("NTL_SKETCH_THRESH", cpw)
        for k, w in ((32, 250), (24, 121), (40, 255), (24, 100), (32, 71), (20, 120)):
            info = {}
            assert pc.check_sketch(dev, seqs, k, w, info=info) > 0
            fb[cpw, w] = info["fallback_strips"]
            assert info["redo_strips"] == 0, (cpw, k, w, info)
    assert fb["4", 250] > 5 * max(fb["10", 250], 1) and fb["0", 250] == 0 and fb["4", 100] > 5 * max(fb["10", 100], 1), fb
    monkeypatch.setenv("NTL_SKETCH_THRESH", "10")
    monkeypatch.setenv("NTL_SKETCH_THRESH_DIRECT", "1")
    assert pc.check_sketch(dev, seqs, 32, 250) > 0
    monkeypatch.setenv("NTL_SKETCH_FORCE_REDO", "1")  # every strip through all three passes
    info = {}
    assert pc.check_sketch(dev, seqs, 32, 250, info=info) > 0
    assert info["fallback_strips"] == info["redo_strips"] > 0


@pytest.mark.parametrize("form", ["fasta", "fasta_wrapped", "fastq", "fastq_wrapped", "fastq_short_quals"])
def test_sim_one_pass_reader(dev, tmp_path, monkeypatch, form):
    """The one-pass reader (ntl_fastx_next_span / _parse_span / _copy_span -> ntl_batch_create_packed_at): every parser thread's
    sequences sit at an upper bound of their place in the packed stream, with gaps between the threads' segments.  Records,
    lengths and the sketches of every batch equal the serial reader's / the oracle's -- N runs, lower case, IUPAC, empty and tiny
    records, ids with comments; wrapped FASTQ and qualities shorter than their bases fall back to the two-pass reader."""
    import oracle
    from ntlink_amd import seqio
    rng = np.random.default_rng(21)
    acgt = np.frombuffer(b"ACGTacgtNR", np.uint8)
    recs = []
    for i in range(160):
        n = int(rng.integers(0, 5000)) if i % 7 else int(rng.integers(0, 40))
        recs.append((f"r{i}", bytes(acgt[rng.choice(10, n, p=[0.23, 0.23, 0.23, 0.23, 0.02, 0.02, 0.01, 0.01, 0.015, 0.005])])))
    p = tmp_path / ("x.fq" if form.startswith("fastq") else "x.fa")
    with open(p, "wb") as fh:
        for name, sq in recs:
            if form == "fasta":
                fh.write(b">" + name.encode() + b" some comment\n" + sq + b"\n")
            elif form == "fasta_wrapped":
                fh.write(b">" + name.encode() + b"\n" + b"\n".join(sq[j:j + 70] for j in range(0, len(sq), 70)) + b"\n")
            elif form == "fastq":
                fh.write(b"@" + name.encode() + b" c\n" + sq + b"\n+\n" + bytes(rng.choice(np.frombuffer(b"@+>I5", np.uint8), len(sq))) + b"\n")
            elif form == "fastq_wrapped":
                q = bytes(rng.choice(np.frombuffer(b"@+>I5", np.uint8), len(sq)))
                fh.write(b"@" + name.encode() + b"\n" + b"\n".join(sq[j:j + 60] for j in range(0, len(sq), 60)) + b"\n+\n" +
                         b"\n".join(q[j:j + 60] for j in range(0, len(q), 60)) + b"\n")
            else:  # one quality line, shorter than the bases: the reader takes "at least one line", so bases > bytes / 2 here
                fh.write(b"@" + name.encode() + b"\n" + sq + b"\n+\nII\n")
    monkeypatch.setenv("NTL_IO_THREADS", "5")
    monkeypatch.setenv("NTL_IO_MIN_CHUNK", "9000")
    want = [(n, s) for n, s in seqio.read_fastx(str(p))]
    if form != "fastq_short_quals":  # (there the reference's reader semantics swallow the following lines as quality: `want` is what counts)
        assert [n for n, _ in want] == [n for n, _ in recs] and [s for _, s in want] == [s for _, s in recs]
    st = {}
    got_names, got_lens, n_one = [], [], 0
    at = 0
    for ss in seqio.load([str(p)], max_bases=60_000, packed=True, stats=st):
        names = ss.names.tolist()
        got_names += names
        got_lens += ss.lengths.tolist()
        n_one += ss.positions is not None
        seqs = [s for _, s in want[at:at + len(names)]]
        at += len(names)
        for k, w in ((24, 20), (12, 64)):
            with dev.batch_packed(ss) as b, dev.sketch(b, k, w) as sk:
                off, h, q, sd = sk.download()
            ooff, oh, op, os_ = oracle.sketch_batch(b"".join(seqs), pc.offsets_of(seqs), k, w)
            assert np.array_equal(off, ooff) and np.array_equal(h, oh) and np.array_equal(q, op) and np.array_equal(sd, os_), (form, k, w)
    assert got_names == [n for n, _ in want] and got_lens == [len(s) for _, s in want]
    if form in ("fasta", "fasta_wrapped", "fastq"):
        assert n_one >= 3 and st.get("one_pass_batches", 0) == n_one  # several batches, all read in one pass


def test_sim_strip_lists(dev, monkeypatch):
    """Round 5: per-strip minimizer lists instead of the bitmask (parity_cases.check_strip_lists)."""
    pc.check_strip_lists(dev, monkeypatch)

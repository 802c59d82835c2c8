"""End-to-end drop-in tests on a real MI355X: the reference's command lines, run through bin/, must
leave the reference's files byte for byte (cf. tests/ntlink_pytest.py:182-198 of the reference)."""
import os
import shutil
import subprocess
import sys

import pytest

from helpers import FIXTURES, GEN, REF, TEST7_PAF, read_text

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")


def _stage(tmp_path, *names):
    for n in names:
        shutil.copy(os.path.join(REF, n), tmp_path / n)


def _check_outputs(tmp_path, prefix, tag, gold, flags=""):
    d = os.path.join(GEN, "fixtures")
    full = tag + flags
    assert read_text(str(tmp_path / (prefix + ".verbose_mapping.tsv"))) == read_text(os.path.join(d, full + ".verbose_mapping.tsv"))
    assert read_text(str(tmp_path / (prefix + ".paf"))) == read_text(os.path.join(d, full + ".paf"))
    assert read_text(str(tmp_path / (prefix + ".pairs.tsv"))) == read_text(os.path.join(d, full + ".pairs.tsv"))
    if gold and not flags:
        exp = os.path.join(REF, "expected_outputs", gold + ".z1000")
        assert read_text(str(tmp_path / (prefix + ".pairs.tsv"))) == read_text(exp + ".pairs.tsv")
        got = read_text(str(tmp_path / (prefix + ".n1.scaffold.dot"))).splitlines()
        ref = read_text(exp + ".n1.scaffold.dot").splitlines()
        assert got[:2] == ref[:2] and got[-1] == "}"
        assert sorted(l for l in got if "->" not in l) == sorted(l for l in ref if "->" not in l)
        assert [l for l in got if "->" in l] == [l for l in ref if "->" in l]


@pytest.mark.parametrize("tag,target,reads,k,w,gold", FIXTURES)
def test_ntlink_pair_driver(tmp_path, tag, target, reads, k, w, gold):
    """`ntLink pair target= reads= k= w= paf=True` (tests/ntlink_pytest.py:184)."""
    _stage(tmp_path, target, reads)
    cmd = [sys.executable, os.path.join(BIN, "ntLink"), "pair", "-B", f"target={target}", f"reads={reads}", f"k={k}", f"w={w}",
           "paf=True", "ntlink_pairs_tsv=True"]
    assert subprocess.call(cmd, cwd=tmp_path) == 0
    prefix = f"{target}.k{k}.w{w}.z1000"
    _check_outputs(tmp_path, prefix, tag, gold)
    if gold:
        assert read_text(str(tmp_path / f"{target}.k{k}.w{w}.tsv")) == read_text(os.path.join(REF, "expected_outputs", gold + ".tsv"))
    if tag.startswith("t7"):
        assert set(read_text(str(tmp_path / (prefix + ".paf"))).splitlines()) == TEST7_PAF


def test_bgzf_reads_and_parallel_readers(tmp_path, monkeypatch):
    """Reads as one BGZF (`bgzip`) FASTQ file: members are inflated in parallel and -- with several readers and tiny chunks -- read
    as ranges of members; outputs byte for byte those of the fixture (test 2 of the reference ships its reads as .fq.gz)."""
    import gzip
    from helpers import write_bgzf
    tag, target, k, w = "t2_k32_w100", "scaffolds_2.fa", 32, 100
    _stage(tmp_path, target)
    text = gzip.open(os.path.join(REF, "long_reads_2.fq.gz"), "rb").read()
    write_bgzf(str(tmp_path / "reads.bgz.fq.gz"), text)
    env = dict(os.environ, NTL_IO_READERS="3", NTL_IO_CHUNK_BYTES="200000", NTL_IO_THREADS="5", NTL_IO_MIN_CHUNK="20000")
    cmd = [sys.executable, os.path.join(BIN, "ntLink"), "pair", "-B", f"target={target}", "reads=reads.bgz.fq.gz", f"k={k}", f"w={w}",
           "paf=True", "ntlink_pairs_tsv=True"]
    assert subprocess.call(cmd, cwd=tmp_path, env=env) == 0
    _check_outputs(tmp_path, f"{target}.k{k}.w{w}.z1000", tag, "scaffolds_2.fa.k32.w100")


def test_makefile_pipe_two_operators(tmp_path):
    """The reference's own recipe (ntLink:198-199,221-225) with bin/indexlr and bin/ntlink_pair.py
    in place of btllib's indexlr and the reference's ntlink_pair.py, reads split over two files."""
    tag, target, k, w = "t2_k32_w100", "scaffolds_2.fa", 32, 100
    _stage(tmp_path, target, "long_reads_2.fq.gz")
    env = dict(os.environ, PATH=BIN + os.pathsep + os.environ["PATH"])
    sh = (f"indexlr --long --pos --strand -k {k} -w {w} -t 4 {target} > {target}.k{k}.w{w}.tsv && "
          f"gzip -f -cd long_reads_2.fq.gz | indexlr --long --pos --strand --len -k {k} -w {w} -t 4 - | "
          f"ntlink_pair.py -p out -n 1 -m {target}.k{k}.w{w}.tsv -s {target} -k {k} -a 1 -z 1000 -f 10 -x 0 "
          f"--verbose --pairs --paf -")
    assert subprocess.call(["bash", "-e", "-o", "pipefail", "-c", sh], cwd=tmp_path, env=env) == 0
    _check_outputs(tmp_path, "out", tag, "scaffolds_2.fa.k32.w100")


def test_sensitive_and_repeat_flags(tmp_path):
    tag, target, reads, k, w = "t3_k24_w250", "scaffolds_3.fa", "long_reads_3.fa.gz", 24, 250
    _stage(tmp_path, target, reads)
    for flag, suffix in (("sensitive=True", ".sensitive"), ("repeats=True", ".repeat_filter")):
        cmd = [sys.executable, os.path.join(BIN, "ntLink"), "pair", f"target={target}", f"reads={reads}", f"k={k}", f"w={w}",
               "paf=True", "ntlink_pairs_tsv=True", flag, "prefix=run" + suffix]
        assert subprocess.call(cmd, cwd=tmp_path) == 0
        _check_outputs(tmp_path, "run" + suffix, tag, None, suffix)


def test_checkpoint_file_bypasses_mapping(tmp_path):
    """bin/ntlink_pair.py:565-575: an existing <prefix>.verbose_mapping.tsv switches to the re-tally."""
    target, reads = "scaffolds_4.fa", "long_reads_4.fa.gz"
    _stage(tmp_path, target, reads)
    prefix = f"{target}.k40.w100.z1000"
    shutil.copy(os.path.join(GEN, "fixtures", "t4_k40_w100.verbose_mapping.tsv"), tmp_path / (prefix + ".verbose_mapping.tsv"))
    cmd = [sys.executable, os.path.join(BIN, "ntLink"), "pair", f"target={target}", f"reads={reads}", "k=40", "w=100", "ntlink_pairs_tsv=True"]
    assert subprocess.call(cmd, cwd=tmp_path) == 0
    assert read_text(str(tmp_path / (prefix + ".pairs.tsv"))) == read_text(os.path.join(GEN, "fixtures", "t4_k40_w100.checkpoint.pairs.tsv"))


def test_error_removes_partial_outputs(tmp_path):
    """Non-zero exit and no partial verbose/paf files (bin/ntlink_pair.py:608-613)."""
    _stage(tmp_path, "scaffolds_4.fa")
    (tmp_path / "bad.tsv").write_text("r1\t100\tnot_a_minimizer\n")
    (tmp_path / "c.tsv").write_text("scaf1\t123:5:+\n")
    cmd = [sys.executable, os.path.join(BIN, "ntlink_pair.py"), "-p", "o", "-m", "c.tsv", "-s", "scaffolds_4.fa", "-k", "40", "--verbose", "--paf", "bad.tsv"]
    assert subprocess.call(cmd, cwd=tmp_path, stderr=subprocess.DEVNULL) != 0
    assert not (tmp_path / "o.verbose_mapping.tsv").exists() and not (tmp_path / "o.paf").exists()


def test_indexlr_pos_only_small_k_w(tmp_path):
    """`indexlr --long --pos -k 15 -w 5` of the overlap stage (ntLink:243-251): H:pos tokens."""
    import oracle
    _stage(tmp_path, "scaffolds_4.fa")
    out = subprocess.check_output([sys.executable, os.path.join(BIN, "indexlr"), "--long", "--pos", "-k", "15", "-w", "5", "-t", "4",
                                   "scaffolds_4.fa"], cwd=tmp_path).decode()
    exp = []
    for name, seq in oracle.read_fastx(os.path.join(REF, "scaffolds_4.fa")):
        h, p, s = oracle.sketch_seq(seq, 15, 5)
        exp.append(name + "\t" + " ".join(f"{int(a)}:{int(b)}" for a, b in zip(h, p)) + "\n")
    assert out == "".join(exp)


def test_btllib_compatible_indexlr_class():
    """btllib.Indexlr(path, k, w, LONG_MODE, t) as used by bin/ntlink_patch_gaps.py:417-441 (gap-fill k20 w10)."""
    import oracle
    import ntlink_amd.btllib as btllib
    path = os.path.join(REF, "long_reads_4_top5.fa")
    ref = list(oracle.read_fastx(path))
    with btllib.Indexlr(path, 20, 10, btllib.IndexlrFlag.LONG_MODE, 4) as recs:
        first = recs.read()
        rest = list(recs)
        assert recs.read() is None
    got = [first] + rest
    assert [r.id for r in got] == [n for n, _ in ref] and [r.readlen for r in got] == [len(s) for _, s in ref]
    for r, (_, seq) in zip(got, ref):
        h, p, s = oracle.sketch_seq(seq, 20, 10)
        assert [m.out_hash for m in r.minimizers] == h.tolist() and [m.pos for m in r.minimizers] == p.tolist()
        assert [m.forward for m in r.minimizers] == [bool(v) for v in s]
    got[0].id = "renamed"  # records are mutable in the reference's use


def test_file_to_file_at_scale_equals_oracle(tmp_path, monkeypatch):
    """The fused driver on a scaled C3 workload, file to file -- 90 Mbp assembly, 0.6 Gbases of ONT-like reads spread over
    plain FASTA, FASTQ and gzip files, 13 read batches, two device worker threads, bases packed by the parser threads --
    leaves the text the oracle derives from the same sequences: `.tsv`, `.verbose_mapping.tsv`, `.paf`, `.pairs.tsv` byte for
    byte, `.dot` edge for edge."""
    import gzip
    import numpy as np
    import oracle
    from ntlink_amd import capi, pipeline, synth
    dev = capi.Device(0)
    wl = synth.DeviceWorkload(dev, "C3", scale=0.03, with_reads=False)
    W = wl.W
    k, w = W["k"], W["w"]
    cbuf, coff = wl.contigs.download()
    ctg_names = ["ctg%06d" % i for i in range(len(coff) - 1)]
    with open(tmp_path / "asm.fa", "wb") as fh:
        for i, n in enumerate(ctg_names):
            s = cbuf[int(coff[i]):int(coff[i + 1])].tobytes()
            fh.write(b">" + n.encode() + b" len=%d\n" % len(s) + b"\n".join(s[j:j + 80] for j in range(0, len(s), 80)) + b"\n")
    files, read_names, rbufs, roffs = [], [], [], []
    for f, kind in enumerate(("fa", "fq", "fa.gz", "fa")):
        rb, _ = wl.make_reads(150_000_000, seed=(9, f))
        buf, off = rb.download()
        rb.close()
        name = f"reads{f}.{kind}"
        opener = gzip.open if kind.endswith(".gz") else open
        with opener(tmp_path / name, "wb") as fh:
            for i in range(len(off) - 1):
                s = buf[int(off[i]):int(off[i + 1])].tobytes()
                rn = b"r%d_%d" % (f, i)
                fh.write((b"@" + rn + b" x\n" + s + b"\n+\n" + b"I" * len(s) + b"\n") if kind == "fq" else (b">" + rn + b"\n" + s + b"\n"))
                read_names.append(rn.decode())
        files.append(name)
        rbufs.append(buf[:int(off[-1])]); roffs.append(off)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("NTL_DEVICE_STREAMS", "2")
    st = pipeline.run_pair(dev, "asm.fa", " ".join(files), k=k, w=w, paf=True, pairs_tsv=True, batch_bases=50_000_000)
    wl.close()
    dev.close()
    assert st["reads"] == len(read_names) and st["read_bases"] == sum(int(o[-1]) for o in roffs)
    # the oracle on the same sequences
    rbuf = np.concatenate(rbufs)
    roff = np.concatenate([[0]] + [o[1:].astype(np.int64) + sum(int(x[-1]) for x in roffs[:i]) for i, o in enumerate(roffs)]).astype(np.uint64)
    c_off, ch, cp, cs = oracle.sketch_batch(cbuf, coff, k, w)
    from helpers import contig_ids
    oix = oracle.Index(ch, contig_ids(c_off), cp, cs)
    r_off, rh, rp, rs = oracle.sketch_batch(rbuf, roff, k, w)
    rlen = np.diff(roff).astype(np.uint32)
    ctg_len = np.diff(coff).astype(np.uint32)
    res = oracle.map_reads(oix, ctg_len, r_off, rlen, rh, rp, rs, k=k, z=1000, x=0.0, sensitive=False, repeat_filter=False, threads=0)
    pre = f"asm.fa.k{k}.w{w}.z1000"
    assert read_text(pre + ".verbose_mapping.tsv") == oracle.format_verbose(res, read_names, ctg_names)
    assert read_text(pre + ".paf") == oracle.format_paf(res, read_names, rlen, ctg_names, ctg_len)
    by_name = dict(zip(ctg_names, ctg_len.tolist()))
    pairs = oracle.filter_pairs(oracle.tally_pairs(res, rlen, ctg_names, ctg_len, k), by_name)
    assert len(pairs) > 100 and read_text(pre + ".pairs.tsv") == oracle.format_pairs(pairs)
    head, nodes, edges = oracle.format_dot(pairs, by_name)
    got = read_text(pre + ".n1.scaffold.dot").splitlines(keepends=True)
    assert got[:2] == head and [l for l in got if "->" in l] == edges and set(l for l in got[2:-1] if "->" not in l) == set(nodes)
    recs = [(n, 0, ch[int(c_off[i]):int(c_off[i + 1])], cp[int(c_off[i]):int(c_off[i + 1])], cs[int(c_off[i]):int(c_off[i + 1])])
            for i, n in enumerate(ctg_names)]
    assert read_text(f"asm.fa.k{k}.w{w}.tsv") == oracle.format_indexlr(recs)


def test_three_processes_shard_the_reads(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 3 -m ntlink_amd.dist_pair pair ...` on real HIP contexts (the three ranks share
    this box's one GPU): every rank parses, maps and writes its byte range of the read files; the files are those of one process."""
    import oracle
    target, k, w = "scaffolds_1.fa", 32, 250
    _stage(tmp_path, target)
    recs = list(oracle.read_fastx(os.path.join(REF, "long_reads_1.fa")))
    cuts = [0, len(recs) // 3, len(recs)]
    names = ["a.fa", "b.fa"]
    for (a, b), n in zip(zip(cuts, cuts[1:]), names):
        with open(tmp_path / n, "w") as fh:
            for name, seq in recs[a:b]:
                fh.write(f">{name}\n{seq.decode() if isinstance(seq, bytes) else seq}\n")
    env = dict(os.environ, NTL_DIST_ONE_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", "29581", "-m", "ntlink_amd.dist_pair", "pair", f"target={target}", "reads=" + " ".join(names), f"k={k}", f"w={w}",
           "paf=True", "ntlink_pairs_tsv=True"]
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    assert subprocess.call(cmd, cwd=tmp_path, env=env, timeout=600) == 0
    _check_outputs(tmp_path, f"{target}.k{k}.w{w}.z1000", "t1_k32_w250", "scaffolds_1.fa.k32.w250")
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]

"""Runs a slice of the parity checks on the SIMT-mock build compiled with -fsanitize=address,undefined
(started by tests/test_sanitizers.py with libasan preloaded; GPU AddressSanitizer is not available)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fuzz_cases  # noqa: E402
import parity_cases as pc  # noqa: E402
from ntlink_amd import capi  # noqa: E402
from sim import simlib  # noqa: E402

capi.DEFAULT_LIB = simlib.build(sanitize=True)  # the host-side entry points (reader, parsers, writers, tally) too
dev = capi.Device(0)
pc.check_sketch(dev, pc.edge_sequences(), 32, 100)
pc.check_sketch(dev, fuzz_cases.fuzz_sequences(3, n=6, max_len=2500), 20, 10)
pc.check_sketch(dev, fuzz_cases.fuzz_sequences(4, n=6, max_len=3000), 70, 3)
pc.check_scenario(dev, "syn_sens_repeat")
pc.check_pair_arrays(dev, *fuzz_cases.fuzz_mapping(2, n_reads=40), k=24, z=1000, x=1.2)

# host side of the C ABI: parallel FASTA/FASTQ reader (cuts inside wrapped quality lines included), gzip,
# TSV parser, emitters, pair tally, the fused driver and the two-operator path
import argparse  # noqa: E402
import gzip  # noqa: E402
import tempfile  # noqa: E402

import numpy as np  # noqa: E402
from helpers import REF  # noqa: E402
from ntlink_amd import formats, pipeline, seqio  # noqa: E402
from test_host import _bgzf, _random_fastx, _records  # noqa: E402

tmp = tempfile.mkdtemp(prefix="ntl_asan_")
rng = np.random.default_rng(11)
os.environ.update(NTL_IO_THREADS="5", NTL_IO_MIN_CHUNK="2000")
for fastq, multiline in ((False, True), (True, False), (True, True)):
    path = os.path.join(tmp, f"x{int(fastq)}{int(multiline)}.fx")
    with open(path, "w", newline="") as fh:
        fh.write(_random_fastx(rng, 1500, fastq, multiline))
    want = list(seqio.read_fastx(path))
    assert _records(list(seqio.load([path], max_bases=100_000))) == want
    with open(path, "rb") as fin, gzip.open(path + ".gz", "wb") as fout:
        fout.write(fin.read())
    assert _records([seqio.load_all([path + ".gz"])]) == want
    os.environ["NTL_IO_NO_MMAP"] = "1"
    assert _records(list(seqio.load([path + ".gz"], max_bases=100_000))) == want
    del os.environ["NTL_IO_NO_MMAP"]
    with open(path, "rb") as fin, open(path + ".bgzf.gz", "wb") as fout:
        fout.write(_bgzf(fin.read(), 20000))
    assert _records(list(seqio.load([path + ".bgzf.gz"], max_bases=100_000))) == want
cwd = os.getcwd()
os.chdir(tmp)
import shutil  # noqa: E402
for n in ("scaffolds_4.fa", "long_reads_4_top5.fa"):
    shutil.copy(os.path.join(REF, n), n)
pipeline.run_pair(dev, "scaffolds_4.fa", "long_reads_4_top5.fa", k=40, w=100, paf=True, pairs_tsv=True)
with open("r.tsv", "w") as fh:
    pipeline.run_indexlr(dev, ["long_reads_4_top5.fa"], 40, 100, fh, True)
pipeline.run_ntlink_pair(dev, argparse.Namespace(FILES=["r.tsv"], s="scaffolds_4.fa", m="scaffolds_4.fa.k40.w100.tsv", p="two", n=1, k=40,
                                                 z=1000, a=1, f=10, x=0.0, checkpoint=None, pairs=True, paf=True, sensitive=False,
                                                 repeat_filter=False, verbose=True))
assert open("two.pairs.tsv").read() == open("scaffolds_4.fa.k40.w100.z1000.pairs.tsv").read()
assert open("two.verbose_mapping.tsv").read() == open("scaffolds_4.fa.k40.w100.z1000.verbose_mapping.tsv").read()
assert len(list(formats.read_indexlr("r.tsv", True, 3000))) > 1
os.chdir(cwd)
shutil.rmtree(tmp)
dev.close()
print("SANITIZERS_CLEAN")

"""Row f5: ntlink_amd.liftover (native ntl_liftover) against the imported reference's outputs
(tests/golden/gen_goldens_liftover.py: the four shipped verbose_mapping + trimmed_scafs.agp pairs and seeded AGPs over the
synthetic scenarios)."""
import gzip
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT
from ntlink_amd import liftover

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = json.load(open(os.path.join(GOLD, "gen", "liftover", "cases.json")))


def _mappings_file(case, tmp_path):
    src = os.path.join(GOLD, case["mappings"])
    if not src.endswith(".gz"):
        return src
    dst = tmp_path / "in.verbose_mapping.tsv"
    dst.write_bytes(gzip.open(src, "rb").read())
    return str(dst)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("threads", ["1", "5"])
def test_liftover_equals_reference(case, threads, tmp_path, monkeypatch):
    monkeypatch.setenv("NTL_IO_THREADS", threads)
    monkeypatch.setenv("NTL_IO_MIN_CHUNK", "2000")  # several pieces even on these small files: cuts at read boundaries
    if threads == "5":
        monkeypatch.setenv("NTL_LIFTOVER_BLOCK", "7000")  # ... and several blocks of pieces
    out = tmp_path / "lifted.tsv"
    agp = liftover.read_agp(os.path.join(GOLD, case["agp"]))
    nin, nout = liftover.liftover_mappings(_mappings_file(case, tmp_path), agp, str(out), case["k"])
    exp = gzip.open(os.path.join(GOLD, "gen", "liftover", case["name"] + ".liftover.tsv.gz"), "rb").read()
    assert out.read_bytes() == exp
    assert nout == case["lines"] == exp.count(b"\n") and nin >= nout


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_liftover_on_the_gpu_box(case, tmp_path, monkeypatch):
    """The same 22 imported-reference cases through the hipcc-built library as it is on the GPU box (row f5 is host code, but
    the library the driver's GPU tier loads is the one that must pass), default thread count of that host."""
    from ntlink_amd import capi
    assert capi.load()._name == capi.DEFAULT_LIB
    monkeypatch.setenv("NTL_IO_MIN_CHUNK", "2000")
    out = tmp_path / "lifted.tsv"
    agp = liftover.read_agp(os.path.join(GOLD, case["agp"]))
    nin, nout = liftover.liftover_mappings(_mappings_file(case, tmp_path), agp, str(out), case["k"])
    exp = gzip.open(os.path.join(GOLD, "gen", "liftover", case["name"] + ".liftover.tsv.gz"), "rb").read()
    assert out.read_bytes() == exp
    assert nout == case["lines"] and nin >= nout


def test_liftover_cli_and_errors(tmp_path):
    case = CASES[0]
    out = tmp_path / "o.tsv"
    rc = subprocess.call([sys.executable, os.path.join(ROOT, "bin", "ntlink_liftover_mappings.py"), "-m", _mappings_file(case, tmp_path),
                          "-a", os.path.join(GOLD, case["agp"]), "-o", str(out), "-k", str(case["k"])])
    assert rc == 0
    assert out.read_bytes() == gzip.open(os.path.join(GOLD, "gen", "liftover", case["name"] + ".liftover.tsv.gz"), "rb").read()
    agp = liftover.read_agp(os.path.join(GOLD, case["agp"]))
    # empty input -> empty output; a line without four fields or with a malformed token raises and leaves no file
    empty = tmp_path / "empty.tsv"
    empty.write_text("")
    assert liftover.liftover_mappings(str(empty), agp, str(out), 32) == (0, 0) and out.read_bytes() == b""
    ctg = next(iter(agp))
    for bad in ("read1\t%s\t2\n" % ctg, "read1\t%s\t1\t12:+_7\n" % ctg, "read1\t%s\t1\t12:+_x:+\n" % ctg, "\n"):
        p = tmp_path / "bad.tsv"
        p.write_text(bad)
        with pytest.raises(ValueError):
            liftover.liftover_mappings(str(p), agp, str(out), 32)
        assert not out.exists()
    # a contig that is not in the AGP is not parsed at all (the reference returns before splitting the tokens)
    p = tmp_path / "absent.tsv"
    p.write_text("read1\tnot_in_agp\t1\tgarbage\n")
    assert liftover.liftover_mappings(str(p), agp, str(out), 32) == (1, 0)
    with pytest.raises(ValueError):
        bad_agp = tmp_path / "bad.agp"
        bad_agp.write_text("a\t1\t2\n")
        liftover.read_agp(str(bad_agp))

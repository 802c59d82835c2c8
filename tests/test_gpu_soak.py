"""Volume in the driver's own GPU run.  The near-tie and give-up paths of the 32-bit window passes fire about
once per 2^23 .. 2^29 windows, and the rarer shapes of a HiFi mapping about once per 10^5 reads: only gigabases reach them.  These
tests run the builder's soak programs (tests/gpu_volume_soak.py, tests/gpu_map_soak.py: device-generated reads of the BASELINE
workloads with seeds of their own, every record against the oracle) at sizes that fit two minutes on the GPU box's 16 cores (measured: 4.5 Gbases sketched and compared in 6 s)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, timeout=900):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script), *[str(a) for a in args]], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:]
    return p.stdout


@pytest.mark.parametrize("args", [("C3", 8, 1.5e9, 400), ("C5", 6, 1.5e9, 410), ("C3", 4, 1e9, 420, 32, 150), ("C3", 3, 1e9, 430, 32, 500)],
                         ids=["C3_k32_w250", "C5_k24_w100", "C3_reads_k32_w150", "C3_reads_k32_w500"])
def test_sketch_volume_soak(args):
    """12 Gbases of C3 reads at k32 w250, 9 Gbases of C5 reads at k24 w100, 4 Gbases at k32 w150 -- the three shapes of
    sketch_wave_kernel -- and 3 Gbases at k32 w500 (round 6: the large windows) sketched on the device and by the oracle: every minimizer
    record equal."""
    out = _run("gpu_volume_soak.py", *args)
    assert "volume soak clean" in out, out[-2000:]


@pytest.mark.parametrize("args", [("C5", 4, 1e9, 500), ("C3", 4, 1e9, 510, 0)], ids=["C5_records", "C3_for_map_no_records"])
def test_mapping_volume_soak(args):
    """4 Gbases of C5 reads (HiFi, --sensitive, h = 0.92) mapped with both streams overlapping, and (round 6) 4 Gbases of C3 reads
    (ONT, tags) through the record-less call form of the pair driver and the bench (`records=False`: ntl_sketch_run_for_map):
    every mapping, hit and PAF record equal to the oracle's."""
    out = _run("gpu_map_soak.py", *args)
    assert "soak clean" in out, out[-2000:]

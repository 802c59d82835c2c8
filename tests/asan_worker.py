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

dev = capi.Device(0, lib_path=simlib.build(sanitize=True))
pc.check_sketch(dev, pc.edge_sequences(), 32, 100)
pc.check_sketch(dev, fuzz_cases.fuzz_sequences(3, n=12, max_len=4000), 20, 10)
pc.check_sketch(dev, fuzz_cases.fuzz_sequences(4, n=6, max_len=3000), 70, 3)
pc.check_scenario(dev, "syn_sens_repeat")
pc.check_pair_arrays(dev, *fuzz_cases.fuzz_mapping(2, n_reads=40), k=24, z=1000, x=1.2)
pc.check_full_pipeline(dev, pc.fixture_seqs("scaffolds_4.fa"), pc.fixture_seqs("long_reads_4_top5.fa"), 40, 100, z=1000)
dev.close()
print("SANITIZERS_CLEAN")

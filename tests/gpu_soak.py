#!/usr/bin/env python3
"""Randomised differential soak on the GPU: product vs oracle over random (k, w), adversarial sequences and
random hit lists, for a given number of seconds.  Prints the first failing configuration.
Usage: tests/gpu_soak.py [seconds] [seed0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases  # noqa: E402
import parity_cases as pc  # noqa: E402
from ntlink_amd import capi  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = capi.Device(0)
t0 = time.time()
n_sk = n_map = 0
rng = np.random.default_rng(seed)
while time.time() - t0 < budget:
    seed += 1
    k = int(rng.choice([1, 2, 3, 5, 8, 11, 15, 16, 20, 24, 31, 32, 33, 40, 47, 63, 64, 65, 72, 100]))
    w = int(rng.choice([1, 2, 3, 4, 5, 10, 15, 16, 17, 31, 32, 50, 64, 100, 127, 128, 129, 200, 250, 256, 400, 1000]))
    seqs = fuzz_cases.fuzz_sequences(seed, n=int(rng.integers(5, 80)), max_len=int(rng.choice([300, 3000, 9000, 30000])))
    try:
        pc.check_sketch(dev, seqs, k, w)
    except AssertionError as e:
        print("SKETCH MISMATCH seed", seed, "k", k, "w", w, e)
        sys.exit(1)
    n_sk += 1
    arrs = fuzz_cases.fuzz_mapping(seed, n_ctg=int(rng.integers(2, 60)), n_reads=int(rng.integers(20, 600)))
    kw = dict(k=24, z=int(rng.choice([1, 500, 1000, 5001])), x=float(rng.choice([0.0, 0.0, 0.3, 1.0, 1.5, 7.0])),
              sensitive=bool(rng.integers(0, 2)), repeat_filter=bool(rng.integers(0, 2)))
    try:
        pc.check_pair_arrays(dev, *arrs, **kw)
    except AssertionError as e:
        print("MAP MISMATCH seed", seed, kw, e)
        sys.exit(1)
    n_map += 1
print(f"soak clean: {n_sk} sketch configurations, {n_map} mapping configurations in {time.time() - t0:.0f} s")

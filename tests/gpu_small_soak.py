#!/usr/bin/env python3
"""Randomised differential soak of the small-window pass (sketch_small_kernel<W>, 2 <= w <= 15, round 6) on the GPU: product vs oracle
over random k in 3..100 and every window size, on adversarial sequences (ties, N runs, boundary lengths) and on long random ones
(many full strips: the wavefronts that take the path without per-window checks), for a given number of seconds.
Usage: tests/gpu_small_soak.py [seconds] [seed0]"""
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
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
dev = capi.Device(0)
t0 = time.time()
n = strips = bases = 0
rng = np.random.default_rng(seed)
acgt = np.frombuffer(b"ACGT", np.uint8)
while time.time() - t0 < budget:
    seed += 1
    k = int(rng.integers(3, 101))
    w = int(rng.integers(2, 16))
    seqs = fuzz_cases.fuzz_sequences(seed, n=int(rng.integers(5, 40)), max_len=int(rng.choice([3000, 9000, 30000])))
    for _ in range(int(rng.integers(1, 4))):
        s = acgt[rng.integers(0, 4, int(rng.integers(20_000, 400_000)))].copy()
        if rng.integers(0, 2):  # a few N runs and a low-complexity stretch inside a long sequence: multi-run strips between full ones
            for _ in range(int(rng.integers(1, 5))):
                a = int(rng.integers(0, len(s) - 1)); s[a:a + int(rng.choice([1, 2, 17, 300]))] = ord("N")
            a = int(rng.integers(0, len(s) - 1)); s[a:a + int(rng.integers(10, 5000))] = acgt[rng.integers(0, 4)]
        seqs.append(bytes(bytearray(s)))
    info = {}
    try:
        pc.check_sketch(dev, seqs, k, w, info=info)
    except AssertionError as e:
        print("SKETCH MISMATCH seed", seed, "k", k, "w", w, e)
        sys.exit(1)
    n += 1
    strips += info["strips"]
    bases += sum(len(q) for q in seqs)
print(f"small-window soak clean: {n} configurations, {strips} strips, {bases / 1e6:.0f} Mbases, {time.time() - t0:.0f} s")

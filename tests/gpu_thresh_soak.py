#!/usr/bin/env python3
"""Randomised differential soak of the threshold window pass (sketch_wave_kernel / sketch_thresh_kernel + sketch_fast_list_kernel) on the GPU: product
vs oracle over random k, w in 71..255 and (round 6: the wave kernel's large windows) 256..1135, candidates per window, staged / direct list form, on adversarial and on long random
sequences, for a given number of seconds.  Usage: tests/gpu_thresh_soak.py [seconds] [seed0]"""
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
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
dev = capi.Device(0)
t0 = time.time()
n = fb = strips = 0
rng = np.random.default_rng(seed)
acgt = np.frombuffer(b"ACGT", np.uint8)
while time.time() - t0 < budget:
    seed += 1
    k = int(rng.integers(5, 101))
    w = int(rng.integers(71, 256)) if rng.integers(0, 4) else int(rng.integers(256, 1136))  # one configuration in four: a large window
    cpw = str(rng.choice(["10", "10", "6", "8", "13", "4"]))
    os.environ["NTL_SKETCH_THRESH"] = cpw
    os.environ["NTL_SKETCH_THRESH_DIRECT"] = str(int(rng.integers(0, 2)))
    os.environ["NTL_SKETCH_WAVE"] = str(rng.choice(["1", "1", "4", "16", "0"]))  # sketch_wave_kernel's shapes (k <= 64) / sketch_thresh_kernel
    seqs = fuzz_cases.fuzz_sequences(seed, n=int(rng.integers(5, 40)), max_len=int(rng.choice([3000, 9000, 30000])))
    for _ in range(int(rng.integers(1, 4))):  # long random sequences: full strips, hundreds of candidates each
        seqs.append(bytes(acgt[rng.integers(0, 4, int(rng.integers(20_000, 400_000)))]))
    info = {}
    try:
        pc.check_sketch(dev, seqs, k, w, info=info)
    except AssertionError as e:
        print("SKETCH MISMATCH seed", seed, "k", k, "w", w, "cpw", cpw, "direct", os.environ["NTL_SKETCH_THRESH_DIRECT"], "wave", os.environ["NTL_SKETCH_WAVE"], e)
        sys.exit(1)
    n += 1
    fb += info["fallback_strips"]
    strips += info["strips"]
print(f"threshold-pass soak clean: {n} configurations, {strips} strips, {fb} through the block-minima pass, {time.time() - t0:.0f} s")

#!/usr/bin/env python3
"""Round 6: the sketch of the stages around `pair` (k15 w5, k20 w10, ...) with sketch_small_kernel against the
round-1 forms it replaced (NTL_SKETCH_SMALL=0: four / one k-mer per lane), same process, same batch; one JSON line per (k, w, form):
bench.py's dense_sketch_line (whole sketch call, window pass and emit kernel by the library's own event spans)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ntlink_amd import capi  # noqa: E402


def main():
    cases = [(15, 5), (20, 10), (15, 2), (15, 3), (20, 8), (20, 15)]
    if len(sys.argv) > 1:
        cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    dev = capi.Device(0)
    for k, w in cases:
        for small in ("1", "0"):
            os.environ["NTL_SKETCH_SMALL"] = small
            d = bench.dense_sketch_line(dev, k, w)
            r = d["roofline"]
            print(json.dumps({"k": k, "w": w, "sketch_small_kernel": small == "1", "Gbases_per_s": d["value"], "ms_per_sketch": d["ms_per_sketch"],
                              "minimizers": d["minimizers"], "stage_ms": d["stage_ms"], "window_pass_Gbases_per_s": r["window_kernel"]["Gbases_per_s"],
                              "hbm_frac": r["frac"], "device": dev.name}), flush=True)
    os.environ.pop("NTL_SKETCH_SMALL", None)
    dev.close()


if __name__ == "__main__":
    main()

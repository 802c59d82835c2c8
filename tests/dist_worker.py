"""Worker of tests/test_dist_gloo.py: one rank of `ntlink_amd.dist_pair` on CPU (gloo), with the SIMT-mock
build of the kernels standing in for the GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ntlink_amd import dist_pair  # noqa: E402
from sim import simlib  # noqa: E402

if __name__ == "__main__":
    sys.exit(dist_pair.main(sys.argv[1:], device_factory=lambda local: simlib.device()))

"""Worker of tests/test_dist_gloo.py: one rank of `ntlink_amd.dist_pair` on CPU (gloo); NTLINK_AMD_LIB points the C-ABI
binding at the SIMT-mock build of the kernels, which stands in for the GPUs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ntlink_amd import dist_pair  # noqa: E402

if __name__ == "__main__":
    assert os.environ.get("NTLINK_AMD_LIB"), "the test sets NTLINK_AMD_LIB to the mock build"
    if os.environ.get("NTL_TEST_FAKE_HOSTS"):  # the ranks pretend to sit on that many hosts, dealt out in turn (pipeline.shared_contigs)
        os.environ["NTL_FAKE_HOSTNAME"] = "host%d" % (int(os.environ["RANK"]) % int(os.environ["NTL_TEST_FAKE_HOSTS"]))
    sys.exit(dist_pair.main(sys.argv[1:]))

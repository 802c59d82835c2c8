import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_xdist_auto_num_workers(config):
    """`-n auto` of pytest.ini: one process where a GPU is present (the -m gpu tests share the device and their native library is
    what the run is about), one worker per CPU but one otherwise (NTL_PYTEST_WORKERS overrides; 0 = serial)."""
    if "NTL_PYTEST_WORKERS" in os.environ:
        return int(os.environ["NTL_PYTEST_WORKERS"])
    if os.path.exists("/dev/kfd"):
        return 0
    return max(1, min(8, (os.cpu_count() or 2) - 1))  # 8 CPUs: 7 workers run the suite in 6 min, 4 in 11.6 (the mock's lanes are threads that mostly wait)

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


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite runs the kernels under a SIMT mock (one pthread per lane): minutes of wall time when serial.  Where pytest-xdist is
    loaded and nobody gave a worker count, ask it for `-n auto` here, in front of its own pytest_cmdline_main (NOT in pytest.ini's
    addopts: a box without xdist must still run `pytest -m gpu`, and `-p no:xdist` must work)."""
    if config.pluginmanager.hasplugin("xdist") and getattr(config.option, "numprocesses", None) is None \
            and not hasattr(config, "workerinput"):
        config.option.numprocesses = "auto"


@pytest.hookimpl(optionalhook=True)
def pytest_xdist_auto_num_workers(config):
    """`-n auto` (added above): one process where a GPU is present (the -m gpu tests share the device and their native library is
    what the run is about), one worker per CPU but one otherwise (NTL_PYTEST_WORKERS overrides; 0 = serial)."""
    if "NTL_PYTEST_WORKERS" in os.environ:
        return int(os.environ["NTL_PYTEST_WORKERS"])
    if os.path.exists("/dev/kfd"):
        return 0
    return max(1, min(8, (os.cpu_count() or 2) - 1))  # 8 CPUs: 7 workers run the suite in 6 min, 4 in 11.6 (the mock's lanes are threads that mostly wait)

"""AddressSanitizer + UBSan over the kernels' source (CPU build under the SIMT mock) and the host side of
the C ABI."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib(name):
    p = subprocess.check_output(["gcc", "-print-file-name=" + name]).decode().strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_kernels_clean_under_asan_ubsan():
    asan, ubsan = _lib("libasan.so"), _lib("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("sanitizer runtimes not installed")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               LD_PRELOAD=asan + " " + ubsan)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_worker.py")], env=env, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0 and "SANITIZERS_CLEAN" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])

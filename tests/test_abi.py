"""The C-ABI library loads and exports every symbol include/ntlink_amd.h declares (no compute here)."""
import ctypes
import os
import re

import pytest

from ntlink_amd import capi
from ntlink_amd.build import build_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ntlink_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ntl_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    path = build_hip()
    lib = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} missing from {path}"


def test_no_cpu_fallback():
    """Without a GPU the product refuses to run instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.NtlError):
        capi.Device(0)

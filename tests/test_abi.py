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


def test_library_signature_is_checked(monkeypatch, tmp_path):
    """Round 6: staleness by content.  The built library carries the sha256 of its sources and flags; `capi.load` refuses the default
    library when that signature is not the sources' (objects and the .so travel to the GPU box: what runs must be what is committed),
    and a second `build_hip` reuses everything."""
    from ntlink_amd import build
    build_hip()
    assert build.library_is_current()
    build_hip()
    assert build.last_action == "reused"
    sig = open(build.OUT + ".sig").read().strip()
    assert sig == build.source_signature() and len(sig) == 64
    assert build.source_signature(("-DNTL_SOMETHING",)) != sig  # the flags are part of it
    monkeypatch.setattr(build, "library_is_current", lambda *a: False)
    monkeypatch.setattr(capi, "_libs", {})
    with pytest.raises(capi.NtlError, match="signature"):
        capi.load()

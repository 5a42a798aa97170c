"""The C-ABI library loads on a box without a GPU and exports every entry point that
include/m2d.h declares (no compute calls here), and the ctypes table mirrors the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "m2d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(m2d_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("m2d_conv1d_fwd", "m2d_conv1d_bwd_data", "m2d_conv1d_bwd_weight", "m2d_gemm", "m2d_bn_fwd",
                 "m2d_bn_bwd", "m2d_gru_layer_fwd", "m2d_gru_layer_bwd", "m2d_gp_interpolate", "m2d_gp_penalty_fwd",
                 "m2d_gp_penalty_bwd", "m2d_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from music2dance_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from music2dance_amd import build
        build.build(verbose=False)
    h = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(h, s)]
    assert not missing, "declared in include/m2d.h but not exported: %s" % missing


def test_ctypes_table_matches_header():
    from music2dance_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    h = _lib.lib()
    assert h.m2d_version() >= 100
    # argument counts of the binding equal the header's parameter counts
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "m2d.h")).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), (name, n, len(args))


def test_missing_library_fails_loudly(monkeypatch):
    from music2dance_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libm2d_hip.so")
    with pytest.raises(_lib.M2dError):
        _lib.lib()

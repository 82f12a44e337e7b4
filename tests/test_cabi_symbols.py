"""CPU: libptpp_hip.so loads and exports every symbol include/ptpp.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptpp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptpp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from promptttspp_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run `make -C promptttspp_amd/csrc` (or __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ptpp.h but not exported"
    # and the Python binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == names


def test_binding_loads_and_reports_version():
    from promptttspp_amd import _lib

    lib = _lib.load()
    assert lib.ptpp_version() >= 1
    assert lib.ptpp_conv_cin_padded(80, _lib.BF16) == 96
    assert lib.ptpp_conv_cin_padded(256, _lib.BF16) == 256
    assert lib.ptpp_conv_cin_padded(32, _lib.F32) == 32


def test_bad_arguments_fail_loudly_without_gpu():
    import pytest
    import torch

    from promptttspp_amd import _lib, ops

    with pytest.raises(_lib.PtppError):
        ops.bct_to_btc(torch.zeros(1, 4, 4), torch.float32)  # CPU tensor: no fallback
    a = _lib.ConvArgs()
    assert _lib.load().ptpp_conv1d_fwd(ctypes.byref(a), None) == -1
    assert b"null" in _lib.load().ptpp_last_error()

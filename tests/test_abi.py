"""CPU-side checks of the C ABI: the library loads and exports every symbol that
include/focalsv_hip.h declares; host-only entry points behave (no compute calls here)."""
import os
import re

import numpy as np
import pytest

from focalsv_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "focalsv_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fsv_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/focalsv_hip.h but not exported"


def test_strerror_and_version():
    assert _lib.load().fsv_version() >= 100
    assert "fallback" in _lib.strerror(-1)


def test_pack_reads_layout():
    words, off, lens = _lib.pack_reads(["ACGT" * 5, "T", "GGNCA"])
    assert list(lens) == [20, 1, 5]
    assert list(off) == [0, 2, 3, 4]
    w0 = int(words[0])
    assert [(w0 >> (2 * i)) & 3 for i in range(4)] == [0, 1, 2, 3]
    assert int(words[2]) == 3
    w3 = int(words[3])
    assert [(w3 >> (2 * i)) & 3 for i in range(5)] == [2, 2, 0, 1, 0]  # N -> A


def test_no_device_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.FsvError) as e:
        _lib.Context(0)
    assert e.value.code == -1


def test_struct_sizes_match_header():
    assert _lib.WTASK_DTYPE.itemsize == 32 and _lib.WRES_DTYPE.itemsize == 16

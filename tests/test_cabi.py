"""The C-ABI shared library: it loads without a GPU, exports every symbol that
include/sknnr_hip.h declares, and fails loudly (no CPU fallback) when no device exists."""

from __future__ import annotations

import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "sknnr_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sknnr_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def native():
    from sknnr_amd import _build, _native

    if not os.path.exists(_build.LIB_PATH):
        _build.build()
    _native.load()
    return _native


def test_header_and_binding_agree(native):
    assert declared_functions() == sorted(native.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(native):
    lib = ctypes.CDLL(native.library_path())
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in include/sknnr_hip.h but not exported"


def test_no_other_symbols_leak(native):
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", native.library_path()], capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    exported = {s for s in exported if not s.startswith("_")}
    assert exported == set(declared_functions())


def test_abi_version_and_last_error(native):
    lib = native.load()
    assert lib.sknnr_abi_version() == native.ABI_VERSION == 4
    assert isinstance(lib.sknnr_last_error(), bytes)


def test_struct_layouts_match_the_header(native):
    assert ctypes.sizeof(native.QueryOpts) == 40
    assert native.QueryOpts.row_offset.offset == 32
    assert native.QueryOpts.check_finite.offset == 28
    assert ctypes.sizeof(native.Stats) == 88  # ABI v4: + mfma_executed_ratio
    assert native.Stats.total_kernel_ms.offset == 48


def test_argument_errors_without_touching_a_device(native):
    lib = native.load()
    assert lib.sknnr_index_create(None, 10, 3, None, 0, 0, ctypes.byref(ctypes.c_void_p())) == native.ERR_INVALID
    assert b"ref must be" in lib.sknnr_last_error()
    assert lib.sknnr_kneighbors(None, None, 1, None, None, None, 0, None) == native.ERR_INVALID
    assert lib.sknnr_check_finite(None, None) == native.ERR_INVALID
    assert lib.sknnr_stream_begin(None, None, 0, 0, ctypes.byref(ctypes.c_void_p())) == native.ERR_INVALID
    assert lib.sknnr_stream_push(None, None, 1, None, None, None) == native.ERR_INVALID
    assert lib.sknnr_stream_flush(None) == native.ERR_INVALID
    assert lib.sknnr_stream_end(None, None) == 0  # NULL is allowed
    lib.sknnr_index_destroy(None)  # NULL is allowed


@pytest.mark.skipif("__import__('sknnr_amd')._native.device_count() > 0")
def test_fails_loudly_without_a_gpu(native):
    """No CPU fallback: without a device the product path raises."""
    import sknnr_amd

    x = np.random.default_rng(0).standard_normal((20, 4))
    with pytest.raises(native.HipBackendError, match="no HIP device"):
        native.Index(x)
    with pytest.raises(native.HipBackendError, match="no HIP device"):
        sknnr_amd.RawKNNRegressor(n_neighbors=2).fit(x, x[:, :2])
    with pytest.raises(native.HipBackendError, match="no HIP device"):
        sknnr_amd.EuclideanKNNRegressor(n_neighbors=2).fit(x, x[:, :2])

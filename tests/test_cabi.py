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
    assert lib.sknnr_abi_version() == native.ABI_VERSION == 5
    assert isinstance(lib.sknnr_last_error(), bytes)


def test_struct_layouts_match_the_header(native):
    assert ctypes.sizeof(native.QueryOpts) == 48  # ABI v5: + query_dtype, reserved_
    assert native.QueryOpts.row_offset.offset == 40
    assert native.QueryOpts.check_finite.offset == 28
    assert native.QueryOpts.query_dtype.offset == 32
    # the header's struct, compiled: same size and offsets
    import subprocess, tempfile

    src = '#include <stddef.h>\n#include <stdio.h>\n#include "sknnr_hip.h"\nint main(void){printf("%zu %zu %zu %zu\\n", sizeof(sknnr_query_opts), offsetof(sknnr_query_opts, query_dtype), offsetof(sknnr_query_opts, row_offset), sizeof(sknnr_stats));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "layout.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "layout")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        got = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()
    assert [int(v) for v in got] == [48, 32, 40, 88]
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
    assert lib.sknnr_hamming_distances(None, None, 0, None, 1, None, 0, None) == native.ERR_INVALID
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

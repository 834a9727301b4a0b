"""The C-ABI library builds, loads on a GPU-less host and exports every symbol include/scl_hip.h declares.
No compute calls here (no GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from stanford_compression_library_amd.backend import lib as backend_lib

HEADER = os.path.join(ROOT, "include", "scl_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(scl_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(backend_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return ctypes.CDLL(backend_lib.LIB_PATH)


def test_header_declares_the_path():
    names = declared_functions()
    for coder in ("rans", "tans", "range", "aec"):
        for op in ("encode_batch", "decode_batch", "encode_host", "decode_host", "model_create", "model_destroy"):
            assert f"scl_{coder}_{op}" in names
    assert "scl_streams_compact" in names


def test_library_exports_every_declared_symbol(built_lib):
    missing = [n for n in declared_functions() if not hasattr(built_lib, n)]
    assert not missing, f"declared in scl_hip.h but not exported: {missing}"


def test_binding_matches_header(built_lib):
    """the ctypes signature table covers exactly the header's functions"""
    assert sorted(backend_lib._SIGNATURES) == declared_functions()
    backend_lib.load()  # declares argtypes for all of them


def test_abi_version_and_error_string(built_lib):
    built_lib.scl_abi_version.restype = ctypes.c_int
    assert built_lib.scl_abi_version() == 8
    built_lib.scl_last_error.restype = ctypes.c_char_p
    assert isinstance(built_lib.scl_last_error(), bytes)


def test_parameter_validation_needs_no_gpu(built_lib):
    """model_create rejects what the reference asserts on before touching the device"""
    L = backend_lib.load()
    h = ctypes.c_void_p()
    f = (ctypes.c_uint32 * 3)(3, 0, 2)
    assert L.scl_rans_model_create(f, 3, 1, 1, 32, ctypes.byref(h)) == backend_lib.E_PARAM  # zero frequency
    assert b"zero frequency" in L.scl_last_error()
    g = (ctypes.c_uint32 * 3)(3, 3, 3)
    assert L.scl_tans_model_create(g, 3, 1, 32, ctypes.byref(h)) == backend_lib.E_PARAM  # M not a power of two
    big = (ctypes.c_uint32 * 2)(1, 65536)
    assert L.scl_range_model_create(big, 2, 32, 32, ctypes.byref(h)) == backend_lib.E_PARAM  # total > BOTTOM
    assert L.scl_range_model_create(g, 3, 20, 32, ctypes.byref(h)) == backend_lib.E_PARAM  # PRECISION % 8
    assert L.scl_rans_model_create(g, 3, 1 << 62, 8, 32, ctypes.byref(h)) == backend_lib.E_PARAM  # H >= 2^63


def test_product_path_fails_loudly_without_gpu():
    """no CPU fallback: on a host without an MI355X the drop-in classes raise instead of computing"""
    if backend_lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    from stanford_compression_library_amd.compressors.rANS import rANSEncoder, rANSParams
    from stanford_compression_library_amd.core.data_block import DataBlock
    from stanford_compression_library_amd.core.prob_dist import Frequencies

    enc = rANSEncoder(rANSParams(Frequencies({"A": 1, "B": 3})))
    with pytest.raises(backend_lib.SclHipError):
        enc.encode_block(DataBlock(["A", "B", "B"]))

"""CPU: the C-ABI library loads without a GPU, exports every symbol include/quant_engine.h
declares, and its host-side argument checks answer before any device work is attempted."""
import ctypes
import os
import re

import pytest

from quantize_amd import capi, loader

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "quant_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qe_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(loader.lib_path()), "run python -m quantize_amd.build"
    assert os.path.exists(loader.module_path())
    assert loader.lib_path().startswith(os.path.join(REPO, "quantize_amd"))


def test_every_declared_symbol_is_exported():
    L = ctypes.CDLL(loader.lib_path())
    declared = _declared_symbols()
    assert sorted(capi.SYMBOLS) == declared
    for name in declared:
        assert hasattr(L, name), name


def test_identity_strings():
    L = capi.lib()
    assert L.qe_target_arch() == b"gfx950"
    assert L.qe_version().startswith(b"quantize_amd")
    # the reference's own messages, verbatim (tpack.cu:13-14)
    assert L.qe_error_string(1) == b"n_bits must be in the range (0, 8]"
    assert L.qe_error_string(2) == b"The input tensor is out of range."
    assert L.qe_error_string(0) == b"ok"


@pytest.mark.parametrize("n,b,expect", [(0, 8, 0), (5, 3, 2), (8, 1, 1), (9, 1, 2), (9408, 4, 4704),
                                        (205520896, 8, 205520896), (2 ** 33, 7, 2 ** 33 * 7 // 8)])
def test_packed_nbytes(n, b, expect):
    assert capi.packed_nbytes(n, b) == expect  # ceil(n*b/8), tpack.cu:224, in 64-bit


def test_host_side_argument_checks_need_no_gpu():
    L = capi.lib()
    # n_bits outside (0, 8] is rejected before any HIP call (CHECK_NBITS, tpack.cu:13)
    assert L.qe_tpack(None, 6, 8, 9, 1, None, None, None) == 1
    assert L.qe_tpack(None, 6, 8, 0, 1, None, None, None) == 1
    assert L.qe_tunpack(None, 8, 12, 0, None, None) == 1
    # n == 0 is a no-op; null pointers with n > 0 are an argument error
    assert L.qe_tpack(None, 6, 0, 8, 1, None, None, None) == 0
    assert L.qe_tpack(None, 6, 8, 8, 1, None, None, None) == 4
    assert L.qe_tpack(ctypes.c_void_p(16), 99, 8, 8, 1, ctypes.c_void_p(16), None, None) == 3  # dtype
    sh = capi.conv_shape(1, 3, 8, 8, 4, 3, 3, 0, 1)  # stride 0
    q = capi.QeQParam(16, 8, 1, 16, 16, 1)
    assert L.qe_quantconv2d(ctypes.byref(q), ctypes.byref(q), None, ctypes.byref(sh), ctypes.c_void_p(16),
                            None, 0, None) == 4
    sh = capi.conv_shape(1, 3, 8, 8, 4, 3, 3, 1, 1)
    bad = capi.QeQParam(16, 9, 1, 16, 16, 1)
    assert L.qe_quantconv2d(ctypes.byref(bad), ctypes.byref(q), None, ctypes.byref(sh), ctypes.c_void_p(16),
                            None, 0, None) == 1

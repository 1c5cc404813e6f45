"""The C-ABI library loads on a CPU-only box and exports every symbol include/saspa_hip.h
declares (no compute calls without a GPU); argument validation happens on the host before
any launch."""
import ctypes as C
import os
import re

import pytest

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "saspa_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(saspa_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} missing: run __graft_entry__.build()")
    lib = _lib.load()
    declared = _declared_symbols()
    assert declared and set(declared) == set(_lib.SYMBOLS), (declared, sorted(_lib.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.saspa_abi_version() == 19 and lib.saspa_build_arch() == b"gfx950"


def test_host_side_argument_validation_needs_no_gpu():
    lib = _lib.load()
    p = _lib.GemmParams()
    assert lib.saspa_gemm(C.byref(p), None) == -1                    # null pointers
    a = (C.c_char * 64)()
    base = C.addressof(a)
    p.a0 = p.w = p.out = (base + 15) // 16 * 16
    p.dtype, p.M, p.N, p.K, p.batch = 0, 16, 16, 12, 1
    p.kh = p.kw = p.stride = 1
    p.c0, p.lda0, p.ldw, p.ldo, p.hin, p.win, p.hout, p.wout = 12, 12, 12, 16, 16, 1, 16, 1
    assert lib.saspa_gemm(C.byref(p), None) == -2                    # channels not a multiple of 8
    p.c0 = p.K = p.lda0 = p.ldw = 16
    p.K = 32
    assert lib.saspa_gemm(C.byref(p), None) == -3                    # K != kh*kw*(c0+c1)
    q = _lib.AttnParams()
    assert lib.saspa_flash_attn_bf16(C.byref(q), None) == -1
    assert lib.saspa_canny(None, None, None, 1, 8, 8, 1, 2, None) == -1
    assert lib.saspa_canny(base, base, (base + 15) // 16 * 16, 1, 100000, 32, 1, 2, None) == -3   # bitmaps exceed LDS and W < 64


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()

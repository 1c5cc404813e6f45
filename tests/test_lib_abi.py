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
    assert lib.saspa_abi_version() == 20 and lib.saspa_build_arch() == b"gfx950"


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
    # ABI 18 deferred reduce without K slices (no workspace): nothing would write the slabs the consumer sums -> refused
    p.K = p.c0 = p.lda0 = p.ldw = 16
    p.nb1 = p.nb2 = 1
    p.alpha = 1.0
    assert lib.saspa_gemm_suggest_ksplit(C.byref(p)) == 1
    p.defer_reduce, p.ksplit, p.workspace = 1, 1, None
    assert lib.saspa_gemm(C.byref(p), None) == -3
    p.defer_reduce = 0
    # ABI 19: the halo conv's eligibility and K-slice arithmetic are host-side
    h = _lib.GemmParams()
    h.dtype, h.kh, h.kw, h.stride, h.pad, h.batch, h.hin, h.win, h.hout, h.wout = 0, 3, 3, 1, 1, 2, 16, 16, 16, 16
    h.c0, h.c1, h.N, h.K, h.M, h.ldw, h.lda0, h.ldo, h.korder, h.nb1, h.nb2 = 320, 0, 640, 2880, 512, 2880, 320, 640, 2, 1, 1
    assert lib.saspa_conv3x3_halo_eligible(C.byref(h), None) == 1
    assert lib.saspa_conv3x3_halo_ksplit(C.byref(h), 3) == 3 and lib.saspa_conv3x3_halo_ksplit(C.byref(h), 4) == 3   # 5 chunk pairs: 2 + 2 + 1
    # the `defer_reduce` contract of the halo conv (advisor, round 5): a deferred reduce that would end on ONE slice, or on fewer
    # slices than the caller's saspa_splitk_groupnorm will sum, is refused before anything is launched
    big = (C.c_char * 64)()
    ptr = (C.addressof(big) + 15) // 16 * 16
    h.a0 = h.w = h.out = h.workspace = ptr
    h.defer_reduce, h.ksplit = 1, 1
    assert lib.saspa_conv3x3_halo(C.byref(h), None, None) == -3
    h.ksplit = 4                                                                                                   # lands on 3 slices
    assert lib.saspa_conv3x3_halo(C.byref(h), None, None) == -3
    h.a0 = h.w = h.out = h.workspace = None
    h.defer_reduce, h.ksplit = 0, 1
    h.korder = 1
    assert lib.saspa_conv3x3_halo_eligible(C.byref(h), None) == 0                                                  # im2col packing
    h.korder, h.hin, h.hout = 2, 8, 8
    h.M = 2 * 8 * 16
    assert lib.saspa_conv3x3_halo_eligible(C.byref(h), None) == 0                                                  # 128 pixels per image
    assert lib.saspa_conv3x3_halo(C.byref(h), None, None) == -1                                                    # null operands
    q = _lib.AttnParams()
    assert lib.saspa_flash_attn_bf16(C.byref(q), None) == -1
    assert lib.saspa_canny(None, None, None, 1, 8, 8, 1, 2, None) == -1
    assert lib.saspa_canny(base, base, (base + 15) // 16 * 16, 1, 100000, 32, 1, 2, None) == -3   # bitmaps exceed LDS and W < 64


def test_as_auto_is_the_dispatch_predicate_not_mere_eligibility():
    """ABI 20 (advisor, round 5): `saspa_gemm_as_eligible` says 2 for every size the balanced A-stationary launch can take, but AUTO
    keeps its measured rule -- a ragged block count (352 blocks of 256 rows: 512x704) without LayerNorm / residual and with 320
    columns stays on the tiled kernel.  Callers that plan around the choice (ops.conv drops the epilogue GroupNorm statistics)
    ask `saspa_gemm_as_auto`, the predicate dispatch() itself uses."""
    lib = _lib.load()
    keep = []

    def mk(m, n=320, res=False, act=0):
        p = _lib.GemmParams()
        a = (C.c_char * 64)()
        keep.append(a)
        base = (C.addressof(a) + 15) // 16 * 16
        p.a0 = p.w = p.out = base
        if res:
            p.residual, p.ldr = base, n
        p.dtype, p.M, p.N, p.K, p.batch = _lib.SASPA_BF16, m, n, 320, 1
        p.kh = p.kw = p.stride = 1
        p.c0 = p.lda0 = p.ldw = 320
        p.ldo, p.hin, p.hout, p.win, p.wout, p.nb1, p.nb2, p.alpha, p.act = n, m, m, 1, 1, 1, 1, 1.0, act
        return p
    for m, n, res, act, elig, auto in [(65536, 320, False, 0, 2, 1), (352 * 256, 320, False, 0, 2, 0), (352 * 256, 320, True, 0, 2, 1),
                                       (352 * 256, 640, False, 0, 2, 1), (352 * 256, 2560, False, 3, 2, 0), (100 * 256, 320, False, 0, 0, 0)]:
        p = mk(m, n, res, act)
        assert lib.saspa_gemm_as_eligible(C.byref(p)) == elig, (m, n, res)
        assert lib.saspa_gemm_as_auto(C.byref(p)) == auto, (m, n, res)
    p = mk(65536)
    p.variant = 1                                   # pinned to the tiled kernel: AUTO's choice does not apply
    assert lib.saspa_gemm_as_auto(C.byref(p)) == 0


def test_gemm_which_is_a_dry_dispatch():
    """ABI 20: saspa_gemm_which runs saspa_gemm's validation and dispatch without launching anything (no GPU needed) and reports
    family | (ksplit << 8): the level-0 3x3 conv goes to the 8-wave kernel, a K = 320 pointwise layer with a residual to the
    A-stationary one, (16384, 640, 640) to the 4-wave tiles, (4096, 1280, 1280) to the wave-specialised kernel; invalid
    problems return saspa_gemm's own error code."""
    lib = _lib.load()
    a = (C.c_char * 64)()
    base = (C.addressof(a) + 15) // 16 * 16

    def conv(b, h, w, cin, n, kh=3, res=False):
        p = _lib.GemmParams()
        p.a0 = p.w = p.out = base
        p.dtype, p.batch, p.hin, p.hout, p.win, p.wout, p.kh, p.kw, p.stride, p.pad = _lib.SASPA_BF16, b, h, h, w, w, kh, kh, 1, kh // 2
        p.c0 = p.lda0 = cin
        p.K = p.ldw = kh * kh * cin
        p.N = p.ldo = n
        p.M, p.nb1, p.nb2, p.alpha = b * h * w, 1, 1, 1.0
        if res:
            p.residual, p.ldr = base, n
        return p
    for args, fam in [((16, 64, 64, 320, 320), 2), ((16, 64, 64, 320, 320, 1, True), 4), ((16, 32, 32, 640, 640, 1), 1), ((16, 16, 16, 1280, 1280, 1), 3)]:
        w = lib.saspa_gemm_which(C.byref(conv(*args)))
        assert (w & 0xff, w >> 8) == (fam, 1), (args, w)
    bad = conv(16, 64, 64, 320, 320)
    bad.K = 99
    assert lib.saspa_gemm_which(C.byref(bad)) == -3
    assert lib.saspa_gemm_which(None) == -1


def test_ff_block_host_side_validation():
    """saspa_ff_block (ABI 20): null operands, geometry (rows % 128, inner width % 32, pitches) and alignment are refused on the host."""
    lib = _lib.load()
    a = (C.c_char * 64)()
    base = (C.addressof(a) + 15) // 16 * 16
    p = _lib.FfBlockParams()
    assert lib.saspa_ff_block(C.byref(p), None) == -1
    p.x = p.residual = p.out = p.w1 = p.b1 = p.w2f = p.b2 = base
    p.ldx = p.ldr = p.ldo = p.ldw1 = 320
    p.M, p.F = 100, 1280
    assert lib.saspa_ff_block_eligible(C.byref(p)) == 0 and lib.saspa_ff_block(C.byref(p), None) == -3      # rows % 128
    p.M, p.F = 256, 1000
    assert lib.saspa_ff_block_eligible(C.byref(p)) == 0                                                      # inner width % 32
    p.F, p.ldo = 1280, 300
    assert lib.saspa_ff_block_eligible(C.byref(p)) == 0                                                      # pitch < 320
    p.ldo = 320
    assert lib.saspa_ff_block_eligible(C.byref(p)) == 1
    p.ln_gamma = base
    assert lib.saspa_ff_block(C.byref(p), None) == -1                                                        # gamma without beta
    p.ln_beta = base
    p.w2f = base + 8
    assert lib.saspa_ff_block(C.byref(p), None) == -2                                                        # alignment


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()

"""A-stationary GEMM for K = 320 pointwise layers (saspa_gemm_as.hip, ABI 13): plain / bias / residual, fused GEGLU, fused
LayerNorm, transposed tail columns (Q | K | V^T out of one launch) -- against the tiled kernels (bit-equal where the arithmetic
is the same: fp32 accumulation of the same bf16 products in the same K order is NOT guaranteed across kernels, so the bound is
one bf16 ulp of the output scale) and against fp64 references."""
import math

import pytest
import torch

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
from saspa_aug_amd import weights as W

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
K = 320


def _mk(m, n, seed, dev, bias=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m, K, generator=g).to(dev, BF)
    w = (torch.randn(n, K, generator=g) / math.sqrt(K)).to(dev, BF)
    b = torch.randn(n, generator=g).to(dev) if bias else None
    return x, w, b


def _ref(x, w, b, res=None):
    r = x.double().cpu() @ w.double().cpu().t()
    if b is not None:
        r = r + b.double().cpu()
    if res is not None:
        r = r + res.double().cpu()
    return r


# 90 112 rows = 352 row blocks (512x704): since round 5 the steps are dealt evenly, a workgroup runs 1.375 blocks' worth in two or
# three runs; 49 152 = 192 blocks: three quarters of a block each
@pytest.mark.parametrize("m,n", [(49152, 320), (49152, 640), (49152 + 200, 320), (90112, 64), (90112, 320), (90112 + 72, 960)])
@pytest.mark.parametrize("bias,residual", [(True, False), (False, False), (True, True)])
def test_plain(dev, m, n, bias, residual):
    x, w, b = _mk(m, n, m + n, dev, bias)
    res = torch.randn(m, n, generator=torch.Generator().manual_seed(5)).to(dev, BF) if residual else None
    assert ops.linear_ln_fusable(x, w)
    got = ops.linear(x, w, b, residual=res, variant=ops.GEMM_AS)
    auto = ops.linear(x, w, b, residual=res)
    # AUTO takes the kernel where it measured faster (saspa_gemm.hip dispatch()): whole rounds of 256-row blocks (none of the
    # sizes here: 192 / 193 / 352 blocks), or a residual / >= 640 columns
    if residual or n >= 640:
        assert torch.equal(got, auto), "AUTO did not take the A-stationary kernel"
    tiled = ops.linear(x, w, b, residual=res, variant=ops.GEMM_TILED)
    ref = _ref(x, w, b, res)
    scale = ref.abs().max().item()
    err = (got.double().cpu() - ref).abs().max().item()
    # with a residual the product is rounded to bf16 before the add (two roundings, as Linear -> add in the reference and in the
    # tiled kernels' staged epilogue): up to one output ulp
    assert err <= (2 ** -7 if residual else 2 ** -8) * scale, (err, scale)
    d = (got.float() - tiled.float()).abs()
    assert d.max().item() <= 2 ** -7 * scale and (d > 0).float().mean().item() < 2e-2, (d.max().item(), (d > 0).float().mean().item())


def test_not_eligible_falls_back_or_refuses(dev):
    x, w, b = _mk(4096, 320, 1, dev)                        # 16 row blocks: too few for one-workgroup-per-CU
    assert not ops.linear_ln_fusable(x, w)
    ops.linear(x, w, b)                                     # AUTO: the tiled kernels
    with pytest.raises(RuntimeError):
        ops.linear(x, w, b, variant=ops.GEMM_AS)
    g, be = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    with pytest.raises(RuntimeError):
        ops.linear(x, w, b, ln=(g, be, 1e-5))               # a fused LayerNorm needs the kernel


@pytest.mark.parametrize("m,n", [(49152, 2560), (65536, 1280), (49152, 1024), (90112, 2560)])      # 1024: the 128-column GEGLU packing
def test_geglu(dev, m, n):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(m, K, generator=g).to(dev, BF)
    w32 = torch.randn(n, K, generator=g) / math.sqrt(K)
    b32 = torch.randn(n, generator=g)
    wp, bp = W.pack_geglu(w32, b32)
    wp, bp = wp.to(dev, BF), bp.to(dev)
    got = ops.linear(x, wp, bp, act=ops.ACT_GEGLU, variant=ops.GEMM_AS)
    tiled = ops.linear(x, wp, bp, act=ops.ACT_GEGLU, variant=ops.GEMM_TILED)
    assert got.shape == (m, n // 2)
    y = x.double().cpu() @ w32.to(BF).double().t() + b32.double()
    ref = y[:, : n // 2] * torch.nn.functional.gelu(y[:, n // 2:])
    scale = ref.abs().max().item()
    assert (got.double().cpu() - ref).abs().max().item() <= 2 ** -7 * scale
    # the tiled epilogue pairs value and gate through a bf16 staging tile; here both stay fp32 until the product is rounded
    d = (got.float() - tiled.float()).abs()
    assert d.max().item() <= 2 ** -6 * scale
    assert (got.double().cpu() - ref).abs().mean().item() <= (tiled.double().cpu() - ref).abs().mean().item() * 1.01


@pytest.mark.parametrize("m,n", [(49152, 320), (65536, 2560), (90112, 2560), (98304, 320)])
def test_fused_layernorm(dev, m, n):
    act = ops.ACT_GEGLU if n == 2560 else ops.ACT_NONE
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(m, K, generator=g) * 2 + 0.7).to(dev, BF)
    gamma, beta = (1 + 0.3 * torch.randn(K, generator=g)).to(dev), (0.2 * torch.randn(K, generator=g)).to(dev)
    w32 = torch.randn(n, K, generator=g) / math.sqrt(K)
    b32 = torch.randn(n, generator=g)
    if act == ops.ACT_GEGLU:
        w32, b32 = W.pack_geglu(w32, b32)
    w, b = w32.to(dev, BF), b32.to(dev)
    fused = ops.linear(x, w, b, act=act, ln=(gamma, beta, 1e-5))
    two = ops.linear(ops.layernorm(x, gamma, beta, 1e-5), w, b, act=act, variant=ops.GEMM_AS)
    # same arithmetic for the normalised operand up to the summation order of the row statistics
    d = (fused.float() - two.float()).abs()
    scale = two.float().abs().max().item()
    assert d.max().item() <= 2 ** -6 * scale and (d > 0).float().mean().item() < 2e-2, (d.max().item(), (d > 0).float().mean().item())


@pytest.mark.parametrize("b,ntok", [(12, 4096), (16, 5632)])      # 5632 tokens: the 64x88 level of 512x704 (352 row blocks)
def test_qkv_one_launch(dev, b, ntok):
    """[to_q; to_k; to_v] (960 x 320) with a fused LayerNorm: Q | K row-major, V^T per sample -- against LayerNorm + the two
    launches of the tiled path (models.project_vt)."""
    from saspa_aug_amd import models
    g = torch.Generator().manual_seed(11)
    h = torch.randn(b, ntok, K, generator=g).to(dev, BF)
    gamma, beta = (1 + 0.3 * torch.randn(K, generator=g)).to(dev), (0.2 * torch.randn(K, generator=g)).to(dev)
    wqk = (torch.randn(640, K, generator=g) / math.sqrt(K)).to(dev, BF)
    wv = (torch.randn(320, K, generator=g) / math.sqrt(K)).to(dev, BF)
    wall = torch.cat([wqk, wv], 0).contiguous()
    vt = torch.empty((b, 320, ntok), device=dev, dtype=BF)
    qk = ops.linear(h, wall, None, ln=(gamma, beta, 1e-5), out_t=vt, n_split=640, rows_per_batch=ntok)
    assert qk.shape == (b, ntok, 640)
    n1 = ops.layernorm(h, gamma, beta, 1e-5)
    qk_ref = ops.linear(n1, wqk, variant=ops.GEMM_TILED)
    vt_ref = models.project_vt(n1, wv, ntok)
    for got, ref in ((qk, qk_ref), (vt, vt_ref)):
        d = (got.float() - ref.float()).abs()
        scale = ref.float().abs().max().item()
        assert d.max().item() <= 2 ** -6 * scale and (d > 0).float().mean().item() < 2e-2, (d.max().item(), (d > 0).float().mean().item())


def test_eligibility_levels(dev):
    """saspa_gemm_as_eligible: 0 = cannot, 2 = can and fills the chip (since round 5 the steps are dealt evenly, so every
    eligible size does; 1 = "can, ragged last round" is only returned with SASPA_GEMM_BALANCE=0)."""
    w = torch.zeros(320, K, device=dev, dtype=BF)
    mk = lambda m: torch.zeros(m, K, device=dev, dtype=BF)
    assert ops.linear_ln_fusable(mk(65536), w) == 2            # 256 blocks: one whole round
    assert ops.linear_ln_fusable(mk(90112), w) == 2            # 352 blocks: 512x704
    assert ops.linear_ln_fusable(mk(131072), w) == 2           # 512 blocks
    assert ops.linear_ln_fusable(mk(16384), w) == 0            # 64 blocks: cannot fill the chip
    assert ops.linear_ln_fusable(mk(65536).float(), w) == 0    # bf16 only
    assert ops.linear_ln_fusable(torch.zeros(65536, 640, device=dev, dtype=BF), torch.zeros(320, 640, device=dev, dtype=BF)) == 0   # K = 320 only
    assert ops.linear_ln_fusable(mk(65536), torch.zeros(328, K, device=dev, dtype=BF)) == 0                                       # N % 64

"""Wave-specialised short-K GEMM (csrc/saspa_gemm_ws.hip; SaspaGemmParams.variant = SASPA_GEMM_WS) against a torch fp32
reference on bf16-rounded operands: linear / 1x1 / 3x3 layers, bias, time-embedding row vector, alpha, SiLU / ReLU, fused
GEGLU, ragged M (tiles that end inside M, workgroups with an uneven tile count), both tile widths (N % 160 == 0 or not),
1 .. 20 K-tiles per tile, and bit-equality with the 4-wave kernel on the same problem (same MFMA order per output)."""
import math

import pytest
import torch
import torch.nn.functional as F

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
from saspa_aug_amd import weights as W
from tests.util import from_nhwc, to_nhwc

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _lin(dev, m, n, k, bias=True, act=ops.ACT_NONE, alpha=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m, k, generator=g).to(BF)
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).to(BF)
    b = torch.randn(n, generator=g) if bias else None
    xd, wd, bd = x.to(dev), w.to(dev), (b.to(dev) if bias else None)
    got = ops.linear(xd, wd, bd, act=act, alpha=alpha, variant=ops.GEMM_WS)
    ref = x.float() @ w.float().t()
    if bias:
        ref = ref + b
    ref = ref * alpha
    if act == ops.ACT_SILU:
        ref = F.silu(ref.to(BF).float())            # the kernel rounds to bf16 before the activation pass
    elif act == ops.ACT_RELU:
        ref = F.relu(ref)
    tiled = ops.linear(xd, wd, bd, act=act, alpha=alpha, variant=ops.GEMM_TILED)
    return got.float().cpu()[:, :n], ref, tiled.float().cpu()[:, :n]


@pytest.mark.parametrize("m,n,k", [(65536, 320, 320), (4096, 1280, 1280), (16384, 640, 640), (33000, 320, 64), (40000, 256, 320),
                                   (32768 + 77, 640, 128), (70000, 384, 1280)])
def test_linear_shapes(dev, m, n, k):
    got, ref, tiled = _lin(dev, m, n, k, seed=m % 97)
    assert (got - ref).abs().max() < 3e-2 * max(1.0, ref.abs().max().item()), (m, n, k)
    assert torch.equal(got, tiled), "WS and the 4-wave kernel run the same MFMA sequence per output: bit-equal expected"


@pytest.mark.parametrize("act", [ops.ACT_SILU, ops.ACT_RELU])
def test_activations_alpha_nobias(dev, act):
    got, ref, tiled = _lin(dev, 40000, 320, 320, bias=False, act=act, alpha=0.75, seed=3)
    assert (got - ref).abs().max() < 3e-2 * max(1.0, ref.abs().max().item())
    assert torch.equal(got, tiled)


@pytest.mark.parametrize("m,k,f", [(65536, 320, 1280), (16384, 640, 2560), (4096 * 9, 1280, 640), (33000, 128, 512)])
def test_fused_geglu(dev, m, k, f):
    g = torch.Generator().manual_seed(k)
    x = torch.randn(m, k, generator=g).to(BF)
    w = torch.randn(2 * f, k, generator=g) / math.sqrt(k)
    b = torch.randn(2 * f, generator=g)
    wp, bp = W.pack_geglu(w, b)
    xd, wd, bd = x.to(dev), wp.to(dev, BF), bp.to(dev)
    got = ops.linear(xd, wd, bd, act=ops.ACT_GEGLU, variant=ops.GEMM_WS).float().cpu()
    tiled = ops.linear(xd, wd, bd, act=ops.ACT_GEGLU, variant=ops.GEMM_TILED).float().cpu()
    y = (x.float() @ w.to(BF).float().t() + b).to(BF).float()
    ref = y[:, :f] * F.gelu(y[:, f:])
    assert got.shape == (m, f)
    assert (got - ref).abs().max() < 4e-2 * max(1.0, ref.abs().max().item())
    assert torch.equal(got, tiled)


def test_conv3x3_rowvec_and_1x1(dev):
    g = torch.Generator().manual_seed(9)
    bsz, c, hh, ww, n = 16, 128, 48, 64, 320
    x = torch.randn(bsz, c, hh, ww, generator=g)
    w = torch.randn(n, c, 3, 3, generator=g) / math.sqrt(9 * c)
    b = torch.randn(n, generator=g)
    rv = torch.randn(bsz, n, generator=g)
    xd = to_nhwc(x, BF, dev)
    wq = w.to(BF).float()
    pk = W.pack_conv(wq).to(dev, BF)
    got = from_nhwc(ops.conv(xd, pk, b.to(dev), kh=3, kw=3, pad=1, rowvec=rv.to(dev), variant=ops.GEMM_WS), n)
    tiled = from_nhwc(ops.conv(xd, pk, b.to(dev), kh=3, kw=3, pad=1, rowvec=rv.to(dev), variant=ops.GEMM_TILED), n)
    ref = F.conv2d(xd.float().cpu().permute(0, 3, 1, 2), wq, b, padding=1) + rv[:, :, None, None]
    assert (got - ref).abs().max() < 3e-2 * max(1.0, ref.abs().max().item())
    assert torch.equal(got, tiled)
    w1 = torch.randn(640, c, 1, 1, generator=g) / math.sqrt(c)
    pk1 = W.pack_conv(w1.to(BF).float()).to(dev, BF)
    got = from_nhwc(ops.conv(xd, pk1, None, variant=ops.GEMM_WS), 640)
    tiled = from_nhwc(ops.conv(xd, pk1, None, variant=ops.GEMM_TILED), 640)
    assert torch.equal(got, tiled)


@pytest.mark.parametrize("act", [ops.ACT_NONE, ops.ACT_SILU, ops.ACT_ADD_RELU])
def test_residual(dev, act):
    g = torch.Generator().manual_seed(11)
    m, n, k = 50000, 320, 640
    x = torch.randn(m, k, generator=g).to(BF).to(dev)
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).to(BF).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    r = torch.randn(m, n, generator=g).to(BF).to(dev)
    got = ops.linear(x, w, b, residual=r, act=act, variant=ops.GEMM_WS).float().cpu()
    tiled = ops.linear(x, w, b, residual=r, act=act, variant=ops.GEMM_TILED).float().cpu()
    y = (x.float().cpu() @ w.float().cpu().t() + b.cpu()).to(BF).float()
    ref = {ops.ACT_NONE: lambda: y + r.float().cpu(), ops.ACT_SILU: lambda: F.silu(y) + r.float().cpu(),
           ops.ACT_ADD_RELU: lambda: F.relu(y + r.float().cpu())}[act]()
    assert (got - ref).abs().max() < 4e-2 * max(1.0, ref.abs().max().item())
    assert torch.equal(got, tiled)


def test_not_eligible_is_refused(dev):
    x = torch.randn(4096, 320, device=dev).to(BF)
    w = torch.randn(320, 320, device=dev).to(BF)
    with pytest.raises(RuntimeError):
        ops.linear(x.float(), w.float(), variant=ops.GEMM_WS)

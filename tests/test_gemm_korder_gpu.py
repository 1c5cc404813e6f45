"""Chunk-major K order (SASPA_KORDER_CHUNK, ABI v4): the kh*kw taps of one channel chunk are consecutive K-tiles, so the
LDS-DMA kernels re-read the same input rows back to back.  Every case must equal the tap-major launch of the same
problem BIT FOR BIT (the K-tiles are the same sets of products, visited in another order -- fp32 accumulation order
differs, so equality is asserted against the CPU reference within tolerance and between orders within rounding) and
the CPU reference within the usual tolerance."""
import math

import pytest
import torch
import torch.nn.functional as F

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
from saspa_aug_amd import weights as W
from tests.util import assert_close, from_nhwc, q, to_nhwc

pytestmark = pytest.mark.gpu
BF, F32 = torch.bfloat16, torch.float32


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _chunk(packed, taps, dtype, dev):
    t = W.to_chunk_major(packed, taps, dtype).to(dev, dtype)
    t.saspa_korder = 1
    return t


CASES = [  # b, h, w, cin, cout, k, stride, upsample
    (2, 16, 16, 64, 64, 3, 1, False),
    (2, 16, 16, 128, 320, 3, 1, False),
    (1, 24, 40, 320, 320, 3, 1, False),      # non-square, 5 chunks x 9 taps = 45 K-tiles
    (2, 16, 16, 128, 128, 3, 2, False),      # Downsample2D
    (2, 8, 8, 128, 128, 3, 1, True),         # Upsample2D (nearest x2 folded into the loader)
    (1, 8, 8, 1280, 1280, 3, 1, False),      # deep level: split-K slabs (K = 11 520, M = 64)
]


@pytest.mark.parametrize("dtype", [BF, F32])
@pytest.mark.parametrize("case", CASES)
def test_chunk_major_conv(dev, dtype, case):
    b, h, w_, cin, cout, k, stride, up = case
    x = q(_rand(b, cin, h, w_, seed=7), dtype)
    wt = q(_rand(cout, cin, k, k, seed=8, scale=1 / math.sqrt(cin * k * k)), dtype)
    bias = _rand(cout, seed=9)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xin, wt, bias, stride=stride, padding=k // 2)
    assert W.chunk_major_ok(k, k, cin, 0, dtype)
    xd = to_nhwc(x, dtype, dev)
    tap = ops.conv(xd, W.pack_conv(wt).to(dev, dtype), bias.to(dev), kh=k, kw=k, stride=stride, pad=k // 2, upsample=up)
    chk = ops.conv(xd, _chunk(W.pack_conv(wt), k * k, dtype, dev), bias.to(dev), kh=k, kw=k, stride=stride, pad=k // 2, upsample=up)
    assert_close(from_nhwc(chk, cout), ref, dtype, what=f"chunk-major conv {case}")
    lim = 0.0625 if dtype == BF else 1e-4
    assert (chk.float() - tap.float()).abs().max().item() <= lim * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("variant", [ops.GEMM_TILED, ops.GEMM_WIDE])
def test_chunk_major_concat_rowvec_residual(dev, variant):
    """Up-block resnet conv1: two sources of different widths; the chunk walk crosses from source 0 to source 1 once."""
    b, h, w_, c0, c1, cout = 3, 16, 16, 128, 64, 320
    x0, x1 = q(_rand(b, c0, h, w_, seed=10), BF), q(_rand(b, c1, h, w_, seed=11), BF)
    wt = q(_rand(cout, c0 + c1, 3, 3, seed=12, scale=0.03), BF)
    bias, rv = _rand(cout, seed=13), _rand(b, cout, seed=14)
    res = q(_rand(b, cout, h, w_, seed=15), BF)
    ref = (F.conv2d(torch.cat([x0, x1], 1), wt, bias, padding=1) + rv[:, :, None, None]) * 0.5 + res
    wd = _chunk(W.pack_conv_split(wt, c0, c0, c1, c1), 9, BF, dev)
    out = ops.conv(to_nhwc(x0, BF, dev), wd, bias.to(dev), kh=3, kw=3, pad=1, x2=to_nhwc(x1, BF, dev), rowvec=rv.to(dev),
                   residual=to_nhwc(res, BF, dev), alpha=0.5, variant=variant)
    assert_close(from_nhwc(out, cout), ref, BF, what=f"chunk-major concat conv (variant {variant})")


def test_chunk_major_wide_level0(dev):
    """The bench shape (UNet level 0, CFG batch): AUTO takes the wide kernel; chunk- and tap-major agree with the reference."""
    b, h, w_, c = 4, 64, 64, 320
    x = q(_rand(b, c, h, w_, seed=21), BF)
    wt = q(_rand(c, c, 3, 3, seed=22, scale=1 / math.sqrt(9 * c)), BF)
    bias = _rand(c, seed=23)
    xd = to_nhwc(x, BF, dev)
    chk = ops.conv(xd, _chunk(W.pack_conv(wt), 9, BF, dev), bias.to(dev), kh=3, kw=3, pad=1, variant=ops.GEMM_WIDE)
    tap = ops.conv(xd, W.pack_conv(wt).to(dev, BF), bias.to(dev), kh=3, kw=3, pad=1, variant=ops.GEMM_WIDE)
    ref = F.conv2d(x[:1], wt, bias, padding=1)
    assert_close(from_nhwc(chk[:1]), ref, BF, what="chunk-major wide level-0 conv")
    assert (chk.float() - tap.float()).abs().max().item() <= 0.0625


def test_chunk_major_needs_whole_chunks(dev):
    x = torch.zeros(1, 8, 8, 32, device=dev, dtype=BF)           # 32 channels: not a whole bf16 K-tile
    wt = torch.zeros(64, 9 * 32, device=dev, dtype=BF)
    with pytest.raises(RuntimeError):
        ops.conv(x, wt, kh=3, kw=3, pad=1, korder=1)
    assert not W.chunk_major_ok(3, 3, 32, 0, BF) and W.chunk_major_ok(3, 3, 32, 0, F32) and not W.chunk_major_ok(1, 1, 64, 0, BF)

"""Parity of the 8-wave 256 x 320/256 GEMM variant (csrc/saspa_gemm_pp.hip) against the torch-fp32 CPU
reference of the same op, through the C ABI (SaspaGemmParams.variant = SASPA_GEMM_WIDE pins it; the
last tests use shapes that the library routes there by itself).  bf16 tolerance: tests/util.py."""
import math

import pytest
import torch
import torch.nn.functional as F

from saspa_aug_amd import ops
from saspa_aug_amd import weights as W
from util import assert_close, from_nhwc, q, to_nhwc

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


WIDE_CONV_CASES = [
    # (B, H, W, Cin, Cout, k, stride, upsample)
    (2, 16, 16, 64, 96, 3, 1, False),     # one 256-row tile per image pair, N tail inside the 320-wide tile
    (2, 16, 16, 64, 64, 3, 2, False),     # stride 2 (Downsample2D)
    (1, 8, 12, 64, 48, 3, 1, True),       # nearest x2 + conv (Upsample2D)
    (3, 9, 7, 64, 24, 3, 1, False),       # ragged M (189 rows), halo on every side
    (2, 8, 8, 128, 128, 1, 1, False),     # 1x1
    (2, 24, 24, 128, 256, 3, 1, False),   # N = 256 -> 256x256 tile; M = 1152 = 4.5 tiles
    (1, 16, 16, 512, 320, 3, 1, False),   # K = 4608: the caller's split-K request goes through the wide kernel
    (5, 4, 4, 64, 320, 3, 1, False),      # images of 16 pixels: a tile holds 16 images' time-embedding rows
    (1, 40, 40, 192, 640, 3, 1, False),   # two N tiles, 6.25 M tiles, 27 K-tiles
]


@pytest.mark.parametrize("case", WIDE_CONV_CASES)
def test_wide_conv(dev, case):
    b, h, w_, cin, cout, k, stride, up = case
    x = q(_rand(b, cin, h, w_, seed=7), BF)
    wt = q(_rand(cout, cin, k, k, seed=8, scale=1 / math.sqrt(cin * k * k)), BF)
    bias = _rand(cout, seed=9)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    ref = F.conv2d(xin, wt, bias, stride=stride, padding=k // 2)
    out = ops.conv(to_nhwc(x, BF, dev), W.pack_conv(wt).to(dev, BF), bias.to(dev), kh=k, kw=k, stride=stride, pad=k // 2,
                   upsample=up, variant=ops.GEMM_WIDE)
    assert_close(from_nhwc(out, cout), ref, BF, what=f"wide conv {case}")


@pytest.mark.parametrize("h", [4, 16])
def test_wide_conv_concat_rowvec_residual_silu(dev, h):
    """Up-block resnet conv1 (two sources with different pitches + time-embedding rows), conv2 (residual, alpha),
    and the SiLU epilogue of the cond-embedding convs."""
    b, w_, c0, c1, cout = 3, h, 128, 64, 320
    x0, x1 = q(_rand(b, c0, h, w_, seed=10), BF), q(_rand(b, c1, h, w_, seed=11), BF)
    wt = q(_rand(cout, c0 + c1, 3, 3, seed=12, scale=0.03), BF)
    bias, rv = _rand(cout, seed=13), _rand(b, cout, seed=14)
    res = q(_rand(b, cout, h, w_, seed=15), BF)
    conv = F.conv2d(torch.cat([x0, x1], 1), wt, bias, padding=1) + rv[:, :, None, None]
    wd = W.pack_conv_split(wt, c0, c0, c1, c1).to(dev, BF)
    a0, a1 = to_nhwc(x0, BF, dev), to_nhwc(x1, BF, dev)
    out = ops.conv(a0, wd, bias.to(dev), kh=3, kw=3, pad=1, x2=a1, rowvec=rv.to(dev), residual=to_nhwc(res, BF, dev), alpha=0.5,
                   variant=ops.GEMM_WIDE)
    assert_close(from_nhwc(out, cout), conv * 0.5 + res, BF, what="wide concat conv")
    out = ops.conv(a0, wd, bias.to(dev), kh=3, kw=3, pad=1, x2=a1, rowvec=rv[0].to(dev), act=ops.ACT_SILU, variant=ops.GEMM_WIDE)
    ref = F.silu(F.conv2d(torch.cat([x0, x1], 1), wt, bias, padding=1) + rv[0][None, :, None, None])
    assert_close(from_nhwc(out, cout), ref, BF, what="wide concat conv, shared row, SiLU")


@pytest.mark.parametrize("b,h,w_", [(9, 8, 8), (7, 8, 11), (11, 7, 7), (5, 6, 7)])
def test_wide_conv_rowvec_images_smaller_than_a_half_tile(dev, b, h, w_):
    """Time-embedding rows where a 128-row half of the tile spans several images: 43 <= H*W < 128 reads the rows of up to four
    images from the LDS slots the tile set-up staged (8x8, 8x11: the deepest UNet level at 512x512 / 512x704; 7x7: four images
    per half), H*W < 43 takes the per-fragment global-load path.  Un-split (ksplit = 1), ragged M, residual + alpha."""
    cin, cout = 64, 320
    x = q(_rand(b, cin, h, w_, seed=31), BF)
    wt = q(_rand(cout, cin, 3, 3, seed=32, scale=1 / math.sqrt(9 * cin)), BF)
    bias, rv = _rand(cout, seed=33), _rand(b, cout, seed=34)
    res = q(_rand(b, cout, h, w_, seed=35), BF)
    ref = (F.conv2d(x, wt, bias, padding=1) + rv[:, :, None, None]) * 0.5 + res
    out = ops.conv(to_nhwc(x, BF, dev), W.pack_conv(wt).to(dev, BF), bias.to(dev), kh=3, kw=3, pad=1, rowvec=rv.to(dev),
                   residual=to_nhwc(res, BF, dev), alpha=0.5, variant=ops.GEMM_WIDE, ksplit=1)
    assert_close(from_nhwc(out, cout), ref, BF, what=f"wide conv, rows of several images per half tile ({b}x{h}x{w_})")


@pytest.mark.parametrize("b,h,w_,res_on", [(2, 16, 16, True), (3, 16, 24, False), (1, 32, 32, True)])
def test_wide_conv_epilogue_groupnorm_statistics(dev, b, h, w_, res_on):
    """GroupNorm statistics out of the wide kernel's epilogue (per 128-row block and unit of 10 channels: sum / sum of squares of
    the STORED bf16 values, accumulated over the two 64-row passes of a wave group), with and without a residual."""
    cin, cout, unit = 64, 320, 10
    x = q(_rand(b, cin, h, w_, seed=41), BF)
    wt = q(_rand(cout, cin, 3, 3, seed=42, scale=1 / math.sqrt(9 * cin)), BF)
    bias, rv = _rand(cout, seed=43), _rand(b, cout, seed=44)
    res = q(_rand(b, cout, h, w_, seed=45), BF) if res_on else None
    out = ops.conv(to_nhwc(x, BF, dev), W.pack_conv(wt).to(dev, BF), bias.to(dev), kh=3, kw=3, pad=1, rowvec=rv.to(dev),
                   residual=None if res is None else to_nhwc(res, BF, dev), variant=ops.GEMM_WIDE, ksplit=1, gn_unit=unit)
    ref = F.conv2d(x, wt, bias, padding=1) + rv[:, :, None, None]
    assert_close(from_nhwc(out, cout), ref if res is None else ref + res, BF, what="wide conv with statistics epilogue")
    stats, u = out.saspa_gn[:2]
    assert u == unit
    m = b * h * w_
    stored = out.reshape(m, -1)[:, :cout].float().cpu()
    st = stats.float().cpu()
    assert st.shape == (m // 128, cout // unit, 2)
    for blk in range(m // 128):
        rows = stored[blk * 128:(blk + 1) * 128].reshape(-1, cout // unit, unit)
        torch.testing.assert_close(st[blk, :, 0], rows.sum((0, 2)), rtol=1e-3, atol=2e-2)
        torch.testing.assert_close(st[blk, :, 1], (rows * rows).sum((0, 2)), rtol=1e-3, atol=2e-2)


@pytest.mark.parametrize("m,k,n", [(300, 320, 960), (77, 768, 320), (1, 320, 1280), (4096, 1280, 320), (513, 64, 8)])
def test_wide_linear(dev, m, k, n):
    x, wt, bias = q(_rand(m, k, seed=1), BF), q(_rand(n, k, seed=2, scale=1 / math.sqrt(k)), BF), _rand(n, seed=3)
    res = q(_rand(m, n, seed=4), BF)
    out = ops.linear(x.to(dev, BF), wt.to(dev, BF), bias.to(dev), residual=res.to(dev, BF), variant=ops.GEMM_WIDE)
    assert_close(out.float().cpu(), x @ wt.t() + bias + res, BF, what=f"wide linear {m}x{k}x{n}")


def test_wide_rejects_what_it_cannot_run(dev):
    x = torch.zeros(2, 8, 8, 32, device=dev, dtype=BF)          # 32 channels: a K-tile would straddle taps
    wt = torch.zeros(64, 9 * 32, device=dev, dtype=BF)
    with pytest.raises(RuntimeError):
        ops.conv(x, wt, kh=3, kw=3, pad=1, variant=ops.GEMM_WIDE)
    xf = torch.zeros(64, 64, device=dev)                         # fp32 parity mode has no wide kernel
    with pytest.raises(RuntimeError):
        ops.linear(xf, torch.zeros(64, 64, device=dev), variant=ops.GEMM_WIDE)


def test_auto_dispatch_level0_resnet_conv(dev):
    """The bench shape the variant exists for (UNet level 0: 64x64 latents, 320 channels, CFG batch): AUTO and the
    pinned 4-wave kernel must agree with the CPU reference on the first and last image (whole tensor vs each other)."""
    b, h, w_, c = 12, 64, 64, 320
    x = q(_rand(b, c, h, w_, seed=21), BF)
    wt = q(_rand(c, c, 3, 3, seed=22, scale=1 / math.sqrt(9 * c)), BF)
    bias, rv = _rand(c, seed=23), _rand(b, c, seed=24)
    xd, wd = to_nhwc(x, BF, dev), W.pack_conv(wt).to(dev, BF)
    auto = ops.conv(xd, wd, bias.to(dev), kh=3, kw=3, pad=1, rowvec=rv.to(dev))
    wide = ops.conv(xd, wd, bias.to(dev), kh=3, kw=3, pad=1, rowvec=rv.to(dev), variant=ops.GEMM_WIDE)
    tiled = ops.conv(xd, wd, bias.to(dev), kh=3, kw=3, pad=1, rowvec=rv.to(dev), variant=ops.GEMM_TILED)
    assert torch.equal(auto, wide), "AUTO did not take the wide kernel for M = 49152, N = 320, K = 2880"
    for i in (0, b - 1):
        ref = F.conv2d(x[i:i + 1], wt, bias, padding=1) + rv[i][None, :, None, None]
        assert_close(from_nhwc(wide[i:i + 1]), ref, BF, what=f"level-0 conv image {i} (wide)")
        assert_close(from_nhwc(tiled[i:i + 1]), ref, BF, what=f"level-0 conv image {i} (tiled)")
    # the two kernels sum K in the same tile order per output; they may differ by bf16 rounding only
    assert (wide.float() - tiled.float()).abs().max().item() <= 0.0625


LOOP_CASES = [
    # (kind, B, H, W, Cin, Cout, ksplit): every K-loop flavour must give the SAME bits (same MFMA order per accumulator)
    ("conv", 2, 24, 24, 128, 256, 1),     # 256-wide tile instantiation
    ("conv", 1, 40, 40, 192, 640, 1),     # 320-wide tile, 27 K-tiles, ragged M
    ("conv", 1, 16, 16, 512, 320, 2),     # K slices (fp32 slabs + reduce)
    ("up", 1, 8, 12, 64, 48, 1),          # nearest x2 + conv
    ("lin", 300, 1, 1, 64, 320, 1),       # ONE K-tile: prologue and tail only
    ("lin", 300, 1, 1, 128, 320, 1),      # two K-tiles
    ("lin", 4096, 1, 1, 1280, 320, 1),
]


@pytest.mark.parametrize("case", LOOP_CASES)
def test_wide_loop_flavours_bit_identical(dev, monkeypatch, case):
    """SASPA_GEMM_PP_LOOP (read per launch): 2 = two 40-MFMA intervals per K-tile (default since round 5), 0 = four 20-MFMA
    phases (rounds 2 - 4).  The default must equal both bit for bit and the torch reference within the bf16 tolerance.
    (The round-3 one-barrier loop, =1, exists in the diagnostics library only.)"""
    kind, b, h, w_, cin, cout, ks = case
    outs = {}
    if kind == "lin":
        x, wt, bias = q(_rand(b, cin, seed=51), BF), q(_rand(cout, cin, seed=52, scale=1 / math.sqrt(cin)), BF), _rand(cout, seed=53)
        ref = x @ wt.t() + bias
        run = lambda: ops.linear(x.to(dev, BF), wt.to(dev, BF), bias.to(dev), variant=ops.GEMM_WIDE).float().cpu()
    else:
        up = kind == "up"
        x = q(_rand(b, cin, h, w_, seed=54), BF)
        wt = q(_rand(cout, cin, 3, 3, seed=55, scale=1 / math.sqrt(9 * cin)), BF)
        bias = _rand(cout, seed=56)
        ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x, wt, bias, padding=1)
        xd, wd = to_nhwc(x, BF, dev), W.pack_conv(wt).to(dev, BF)
        run = lambda: from_nhwc(ops.conv(xd, wd, bias.to(dev), kh=3, kw=3, pad=1, upsample=up, variant=ops.GEMM_WIDE, ksplit=ks), cout)
    for loop in ("2", "0"):
        monkeypatch.setenv("SASPA_GEMM_PP_LOOP", loop)
        outs[loop] = run()
    monkeypatch.delenv("SASPA_GEMM_PP_LOOP")
    outs["default"] = run()
    assert_close(outs["default"], ref, BF, what=f"wide kernel, default loop {case}")
    for loop in ("2", "0"):
        assert torch.equal(outs["default"], outs[loop]), f"loop flavour {loop} differs from the default on {case}"


@pytest.mark.parametrize("b,h,w_", [(16, 64, 88), (12, 64, 96), (9, 64, 64)])
def test_wide_conv_tile_counts_just_above_a_round(dev, b, h, w_):
    """352 / 288 / 144 row tiles of 256: since round 5 a tile count just above a whole number of rounds of the 256 CUs launches
    ceil(tiles / rounds) workgroups, every one walking the same number of tiles (64x88 is the level-0 size of the 512x704 bucket)."""
    cin, cout = 64, 320
    x = q(_rand(b, cin, h, w_, seed=61), BF)
    wt = q(_rand(cout, cin, 3, 3, seed=62, scale=1 / math.sqrt(9 * cin)), BF)
    bias, rv = _rand(cout, seed=63), _rand(b, cout, seed=64)
    ref = F.conv2d(x, wt, bias, padding=1) + rv[:, :, None, None]
    out = ops.conv(to_nhwc(x, BF, dev), W.pack_conv(wt).to(dev, BF), bias.to(dev), kh=3, kw=3, pad=1, rowvec=rv.to(dev),
                   variant=ops.GEMM_WIDE, ksplit=1)
    assert_close(from_nhwc(out, cout), ref, BF, what=f"wide conv, {b * h * w_ // 256} row tiles")

"""GroupNorm statistics out of the producers' epilogues (SaspaGemmParams.gn_stats, ABI 12): the (sum, sum of squares) per
(128-row block, unit of 10 channels) that a conv / linear launch leaves beside its output, on every kernel that can carry
them (4-wave 128x160 tiles, the 8-wave 256x320 kernel, both split-K reduce paths), against sums taken from the stored
output; and GroupNorm fed by them (plain, and over a channel concat whose group boundaries fall inside a source) against
the two-launch GroupNorm with its own statistics pass."""
import math

import pytest
import torch

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
from saspa_aug_amd import weights as W

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _stats_ref(out, unit):
    """[B,H,W,N] bf16 -> [rows / 128, N / unit, 2] float64 sums of the STORED values."""
    n = out.shape[-1]
    v = out.double().cpu().reshape(-1, 128, n // unit, unit)
    return torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1)


def _check(out, unit=10):
    stats, u = out.saspa_gn[:2]
    assert u == unit and stats.shape == (out.numel() // out.shape[-1] // 128, out.shape[-1] // unit, 2)
    ref = _stats_ref(out, unit)
    got = stats.double().cpu()
    tol = 2e-5 * ref[..., 1].abs().max().item() + 1e-3
    assert (got - ref).abs().max().item() < tol, ((got - ref).abs().max().item(), tol)


@pytest.mark.parametrize("case", ["tile160", "tile160_res_rowvec", "wide", "wide_splitk", "tile_splitk", "ragged_m", "pointwise", "upsample"])
def test_epilogue_statistics(dev, case):
    g = torch.Generator().manual_seed(sum(map(ord, case)))
    kw = dict(kh=3, kw=3, pad=1)
    b, h, w_, cin, cout = 2, 16, 16, 64, 320
    variant, ksplit, residual, rowvec, up = 0, None, False, False, False
    if case == "tile160_res_rowvec":
        residual, rowvec, cout = True, True, 640
    elif case == "wide":
        b, h, w_, cin, variant = 4, 64, 64, 128, ops.GEMM_WIDE
    elif case == "wide_splitk":
        b, h, w_, cin, cout, variant, ksplit = 2, 32, 32, 640, 640, ops.GEMM_WIDE, 3
    elif case == "tile_splitk":
        cin, cout, variant, ksplit = 1280, 1280, ops.GEMM_TILED, 4
    elif case == "ragged_m":
        b, h, w_, variant = 3, 16, 24, ops.GEMM_WIDE            # M = 1152 = 4.5 wide tiles; 9 row blocks
        cin = 128
    elif case == "pointwise":
        kw = {}
        cin, cout, residual = 320, 320, True
    elif case == "upsample":
        up, b, h, w_, cin, cout = True, 2, 8, 16, 128, 320
    taps = 9 if kw else 1
    x = (torch.randn(b, h, w_, cin, generator=g)).to(dev, BF)
    wt32 = torch.randn(cout, taps * cin, generator=g) / math.sqrt(taps * cin)
    if taps == 9 and cin % 64 == 0:
        wt = W.to_chunk_major(wt32, 9, BF).to(dev, BF)
        wt.saspa_korder = 1
    else:
        wt = wt32.to(dev, BF)
    bias = torch.randn(cout, generator=g).to(dev)
    ho, wo = (2 * h, 2 * w_) if up else (h, w_)
    res = torch.randn(b, ho, wo, cout, generator=g).to(dev, BF) if residual else None
    rv = torch.randn(b, cout, generator=g).to(dev) if rowvec else None
    args = dict(residual=res, rowvec=rv, upsample=up, variant=variant, ksplit=ksplit, **kw)
    out = ops.conv(x, wt, bias, gn_unit=10, **args)
    assert hasattr(out, "saspa_gn"), "the producer did not leave statistics"
    _check(out)
    plain = ops.conv(x, wt, bias, **args)
    assert torch.equal(out, plain), "asking for statistics changed the output"


def test_shapes_that_cannot_carry_statistics_fall_back(dev):
    """hw % 128 != 0 (the 8x8 level: 64 pixels per image) and fp32 outputs: no statistics, no error."""
    x = torch.randn(2, 8, 8, 64).to(dev, BF)
    wt = (torch.randn(320, 64) / 8).to(dev, BF)
    assert not hasattr(ops.conv(x, wt, gn_unit=10), "saspa_gn")
    x32 = torch.randn(2, 16, 16, 64).to(dev)
    assert not hasattr(ops.conv(x32, (torch.randn(320, 64) / 8).to(dev), gn_unit=10), "saspa_gn")


def _producer(dev, b, h, w_, c, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, h, w_, 64, generator=g).to(dev, BF)
    wt = (torch.randn(c, 64, generator=g) / 8).to(dev, BF)
    bias = (torch.randn(c, generator=g) * 2).to(dev)                 # a mean far from zero: E[x^2] - mean^2 must not cancel
    return ops.conv(x, wt, bias, gn_unit=10)


@pytest.mark.parametrize("c0,c1", [(320, 0), (640, 0), (1280, 0), (320, 320), (640, 320), (1280, 640), (1280, 1280)])
@pytest.mark.parametrize("act", [ops.ACT_NONE, ops.ACT_SILU])
def test_groupnorm_from_epilogue_statistics(dev, c0, c1, act):
    """(640 + 320) / 32 = 30 and (1280 + 640) / 32 = 60 channels per group: a group straddles the two sources."""
    b, h, w_ = 2, 16, 24                                               # 384 pixels = 3 row blocks per image
    x = _producer(dev, b, h, w_, c0, 1)
    x2 = _producer(dev, b, h, w_, c1, 2) if c1 else None
    c = c0 + c1
    g = torch.Generator().manual_seed(3)
    gamma, beta = (1 + 0.2 * torch.randn(c, generator=g)).to(dev), (0.3 * torch.randn(c, generator=g)).to(dev)
    fused = ops.groupnorm(x, gamma, beta, 32, 1e-5, act, x2=x2)
    # the same tensors without the attribute -> the two-launch path
    xp, x2p = x.clone(), (x2.clone() if x2 is not None else None)
    assert not hasattr(xp, "saspa_gn")
    plain = ops.groupnorm(xp, gamma, beta, 32, 1e-5, act, x2=x2p)
    d = (fused.float() - plain.float()).abs()
    scale = plain.float().abs().max().item()
    # statistics differ only in summation order (fp32 partials, fp64 combine): outputs equal up to a rare bf16 rounding flip
    assert d.max().item() <= 2 ** -7 * scale and (d > 0).float().mean().item() < 2e-3, (d.max().item(), (d > 0).float().mean().item())
    # and against an fp64 reference of the op
    xc = torch.cat([x, x2], -1) if x2 is not None else x
    v = xc.double().cpu().reshape(b, h * w_, 32, c // 32)
    mean, var = v.mean((1, 3), keepdim=True), v.var((1, 3), keepdim=True, unbiased=False)
    ref = ((v - mean) / torch.sqrt(var + 1e-5)).reshape(b, h, w_, c) * gamma.double().cpu() + beta.double().cpu()
    if act == ops.ACT_SILU:
        ref = ref * torch.sigmoid(ref)
    err = (fused.double().cpu() - ref).abs().max().item()
    assert err < 2.5e-2 * max(1.0, ref.abs().max().item()), err


def test_network_level_equivalence(dev, monkeypatch):
    """A two-level full-width UNet + ControlNet evaluation (320 / 640 channels, 32x32 latents) with and without the epilogue
    statistics (SASPA_GN_FUSE): same result up to bf16 rounding noise; the fused run launches statistics kernels only for
    the GroupNorms whose producers cannot carry them."""
    from saspa_aug_amd import config as CFG
    from saspa_aug_amd import models
    ucfg = dict(CFG.SD15_UNET, block_out=(320, 640), attn=(True, False), layers=1)
    ccfg = dict(ucfg, cond_channels=3, cond_embed=(16, 32, 96, 256))
    sd_u, sd_c = W.synth_state_dict("unet", ucfg, 1), W.synth_state_dict("controlnet", ccfg, 2)
    outs, nstats = {}, {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SASPA_GN_FUSE", mode)
        unet = models.UNet(dict(sd_u), ucfg, dev, BF)
        cn = models.ControlNet(dict(sd_c), ccfg, dev, BF)
        g = torch.Generator().manual_seed(5)
        b, h, w_ = 2, 32, 32
        x = torch.nn.functional.pad(torch.randn(b, h, w_, 4, generator=g), (0, 4)).to(dev, BF)
        ctx = torch.randn(b, 77, ucfg["ctx_dim"], generator=g).to(dev, BF)
        cond = torch.rand(b, 8 * h, 8 * w_, 8, generator=g).to(dev, BF)
        for net in (unet, cn):
            net.prepare_context(ctx)
            net.prepare_timesteps([500])
        cemb = cn.cond_embedding(cond)
        calls = [0]
        real = ops.groupnorm

        def counting(xx, *a, **k):
            x2 = k.get("x2")
            if hasattr(xx, "saspa_gn") and (x2 is None or hasattr(x2, "saspa_gn")) and ops.gn_fusion_enabled():
                calls[0] += 1
            return real(xx, *a, **k)
        monkeypatch.setattr(ops, "groupnorm", counting)
        mid, skips = unet.encode(x, 0)
        s2, m2 = cn.forward(x, 0, cemb, 0.75, skips, mid)
        outs[mode] = unet.decode(m2, s2, 0).float().cpu()
        monkeypatch.setattr(ops, "groupnorm", real)
        nstats[mode] = calls[0]
    assert nstats["0"] == 0 and nstats["1"] >= 12, nstats          # every GroupNorm of both networks but the first of each
    d = (outs["1"] - outs["0"]).abs().max().item()
    assert torch.isfinite(outs["1"]).all() and d < 3e-2 * outs["0"].abs().max().item(), d


@pytest.mark.parametrize("dtype", [BF, torch.float32])
@pytest.mark.parametrize("b,h,w_,cin,cout,ks,rv", [(16, 8, 8, 1280, 1280, None, True), (4, 16, 16, 1280, 1280, 2, True), (2, 16, 16, 640, 1280, 4, False),
                                                  (3, 8, 8, 320, 640, 3, True)])
def test_conv_fused_reduce_groupnorm(dev, dtype, b, h, w_, cin, cout, ks, rv, monkeypatch):
    """ops.conv(..., fuse_gn=...) -> saspa_splitk_groupnorm (ABI 18): split-K reduce + bias + time-embedding row + GroupNorm +
    SiLU of a small-level ResnetBlock2D.conv1 -> norm2 in one launch, against the same conv followed by the GroupNorm launches
    (SASPA_SPLITK_GN=0) and against torch."""
    g = torch.Generator().manual_seed(11 + cin + (ks or 0))
    x = torch.randn(b, h, w_, cin, generator=g).to(dev, dtype)
    wt32 = torch.randn(cout, 9 * cin, generator=g) / math.sqrt(9 * cin)
    if dtype == BF:
        wt = W.to_chunk_major(wt32, 9, BF).to(dev, BF)
        wt.saspa_korder = 1
    else:
        wt = wt32.to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    rowvec = torch.randn(cout, generator=g).to(dev) if rv else None
    gamma, beta = (1 + 0.1 * torch.randn(cout, generator=g)).to(dev), (0.1 * torch.randn(cout, generator=g)).to(dev)
    fg = (gamma, beta, 32, 1e-5, ops.ACT_SILU)
    got = ops.conv(x, wt, bias, kh=3, kw=3, pad=1, rowvec=rowvec, ksplit=ks, fuse_gn=fg)
    monkeypatch.setenv("SASPA_SPLITK_GN", "0")
    two = ops.conv(x, wt, bias, kh=3, kw=3, pad=1, rowvec=rowvec, ksplit=ks, fuse_gn=fg)
    scale = two.float().abs().max().item()
    d = (got.float() - two.float()).abs().max().item()
    assert d <= (1.6e-2 if dtype == BF else 2e-5) * scale, (d, scale)
    # torch reference on the same (rounded) operands
    xr = x.float().cpu().permute(0, 3, 1, 2)
    wr = (wt32.to(dtype).float() if dtype == BF else wt32).view(cout, 3, 3, cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xr, wr, bias.cpu(), padding=1)
    if rv:
        ref = ref + rowvec.cpu()[None, :, None, None]
    if dtype == BF:
        ref = ref.to(BF).float()
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(ref, 32, gamma.cpu(), beta.cpu(), 1e-5)).permute(0, 2, 3, 1)
    e = (got.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert e < (3e-2 if dtype == BF else 1e-4), e

"""saspa_conv3x3_halo (ABI 19): the halo-tiled 3x3 conv with the consuming GroupNorm (+ SiLU) applied to its input tile in
LDS -- ResnetBlock2D's norm -> SiLU -> conv in one launch -- against (a) a PyTorch fp32 reference of the same op with the same
rounding points (F.group_norm -> silu -> round to bf16 -> conv2d) and (b) the two launches it replaces (saspa_groupnorm_apply
+ saspa_gemm); plain form against the im2col kernels.  Concat inputs, bias / time-embedding row / residual / epilogue
statistics, K slices (with and without the deferred reduce + GroupNorm), non-square images, tiles that start mid-row."""
import math

import pytest
import torch
import torch.nn.functional as F

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
from saspa_aug_amd import weights as W

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
SILU = ops.ACT_SILU


def _weights(g, cout, cin, dev):
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)
    pk = W.pack_conv(w)                                           # [N, 9 * C] tap-major
    w32 = W.to_chunk32_major(pk).to(dev, BF)
    w64 = W.to_chunk_major(pk, 9, BF).to(dev, BF) if cin % 64 == 0 else pk.to(dev, BF)
    w64.saspa_korder = 1 if cin % 64 == 0 else 0
    return w.to(BF).float(), w32, w64


def _ref_conv(xn, w, bias, rowvec, residual):
    """xn: [B,H,W,C] fp32 (already bf16-rounded), w: [N,C,3,3] fp32 (bf16-rounded) -> [B,H,W,N] fp32."""
    y = F.conv2d(xn.permute(0, 3, 1, 2).double(), w.double(), None, padding=1).permute(0, 2, 3, 1)
    if bias is not None:
        y = y + bias.double().cpu()
    if rowvec is not None:
        y = y + rowvec.double().cpu()[:, None, None, :]
    y = y.float().to(BF).float()                                   # the epilogue's rounding point
    if residual is not None:
        y = (y + residual.float().cpu()).to(BF).float()
    return y


def _ref_gn(x, gamma, beta, groups, eps, silu):
    y = F.group_norm(x.float().cpu().permute(0, 3, 1, 2), groups, gamma.cpu(), beta.cpu(), eps)
    if silu:
        y = F.silu(y)
    return y.permute(0, 2, 3, 1).to(BF).float()                    # rounded to the storage dtype before the product


CASES = {
    #            b   h   w   c0   c1   n    flags
    "l2":       (2, 16, 16, 64,   0, 320, ""),
    "l2_full":  (1, 16, 16, 128,  0, 256, "bias rowvec res"),
    "l1":       (2, 32, 32, 128,  0, 320, "bias"),
    "l0":       (1, 64, 64, 64,   0, 320, "bias rowvec"),
    "concat":   (2, 32, 32, 64,  64, 640, "bias res stats"),
    "concat96": (1, 32, 32, 96,  32, 320, "bias"),
    "wide_n":   (1, 32, 32, 64,   0, 1280, "bias stats"),
    "nonsq":    (1, 16, 48, 64,   0, 320, "bias res"),           # 768 pixels: tiles start mid-row, fragments straddle rows
    "nonsq88":  (1, 32, 88, 64,   0, 320, "bias"),               # W = 88 (the 512x704 bucket): 16-pixel fragments wrap
    "ksplit":   (1, 16, 16, 256,  0, 320, "bias rowvec stats ks2"),
    "ksplit3":  (2, 16, 16, 320,  0, 640, "bias ks3"),           # 5 chunk pairs on 3 slices: 2 + 2 + 1
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("gn", [False, True])
def test_halo_conv_matches_reference(dev, case, gn):
    b, h, w_, c0, c1, n, flags = CASES[case]
    g = torch.Generator().manual_seed(sum(map(ord, case)) + (7 if gn else 0))
    ctot = c0 + c1
    x = (torch.randn(b, h, w_, c0, generator=g) * 1.5 + 0.3).to(dev, BF)
    x2 = (torch.randn(b, h, w_, c1, generator=g) * 0.7 - 0.2).to(dev, BF) if c1 else None
    wf, w32, w64 = _weights(g, n, ctot, dev)
    bias = torch.randn(n, generator=g).to(dev) if "bias" in flags else None
    rv = torch.randn(b, n, generator=g).to(dev) if "rowvec" in flags else None
    res = torch.randn(b, h, w_, n, generator=g).to(dev, BF) if "res" in flags else None
    ks = 2 if "ks2" in flags else 3 if "ks3" in flags else None
    unit = 10 if "stats" in flags else None
    xc = x if x2 is None else torch.cat([x, x2], -1)
    if gn:
        groups, eps = 32, 1e-5
        gamma = (1 + 0.2 * torch.randn(ctot, generator=g)).to(dev)
        beta = (0.3 * torch.randn(ctot, generator=g)).to(dev)
        gb32 = W.pack_gamma_beta32(gamma.cpu(), beta.cpu()).to(dev)
        out = ops.conv_gn(x, (gb32, groups, eps, SILU), w32, bias, x2=x2, rowvec=rv, residual=res, ksplit=ks, gn_unit=unit)
        assert out is not None, "the launch should be eligible"
        xn = _ref_gn(xc, gamma, beta, groups, eps, True)
        # the two launches it replaces
        hn = ops.groupnorm(x, gamma, beta, groups, eps, SILU, x2=x2)
        two = ops.conv(hn, w64, bias, kh=3, kw=3, pad=1, rowvec=rv, residual=res)
    else:
        out = ops.conv_gn(x, None, w32, bias, x2=x2, rowvec=rv, residual=res, ksplit=ks, gn_unit=unit)
        assert out is not None, "the launch should be eligible"
        xn = xc.float().cpu()
        # (a chunk-major im2col packing needs whole 64-channel chunks per source: compare the 96 + 32 case on the concatenated tensor)
        two = ops.conv(xc.contiguous() if (c1 and c0 % 64) else x, w64, bias, kh=3, kw=3, pad=1, x2=None if (c1 and c0 % 64) else x2,
                       rowvec=rv, residual=res)
    ref = _ref_conv(xn, wf, bias, rv, res)
    got = out.float().cpu()[..., :n]
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    # bf16 output rounding (2^-9 relative) on values up to `scale`, plus a rare flip of a normalised INPUT's bf16 rounding
    assert err < 1.2e-2 * scale, (case, gn, err, scale)
    d2 = (got - two.float().cpu()[..., :n]).abs().max().item()
    assert d2 < 1.2e-2 * scale, (case, gn, "vs two launches", d2, scale)
    # rms: the two HIP paths differ by summation order only (and, with gn, by nothing else: same scale / shift arithmetic)
    rms = ((got - two.float().cpu()[..., :n]) ** 2).mean().sqrt().item() / (ref ** 2).mean().sqrt().item()
    assert rms < 2.5e-3, (case, gn, rms)
    if unit:
        stats = out.saspa_gn[0].double().cpu()
        v = out.double().cpu().reshape(-1, 128, n // unit, unit)
        sref = torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1)
        assert (stats - sref).abs().max().item() < 2e-5 * sref[..., 1].abs().max().item() + 1e-3


def test_halo_conv_statistics_pass_form_and_padding(dev):
    """Input without epilogue statistics (fresh tensor): conv_gn launches the statistics pass itself; an input whose border
    pixels are large checks that the zero padding is applied AFTER the normalisation (SiLU(shift) != 0 must not leak in)."""
    g = torch.Generator().manual_seed(5)
    b, h, w_, c, n = 2, 16, 16, 64, 320
    x = torch.randn(b, h, w_, c, generator=g)
    x[:, 0, :, :] += 4.0
    x[:, :, -1, :] -= 3.0
    x = x.to(dev, BF)
    wf, w32, _ = _weights(g, n, c, dev)
    gamma = (1 + 0.2 * torch.randn(c, generator=g)).to(dev)
    beta = (1.0 + 0.3 * torch.randn(c, generator=g)).to(dev)       # SiLU(beta - ...) far from zero
    gb32 = W.pack_gamma_beta32(gamma.cpu(), beta.cpu()).to(dev)
    out = ops.conv_gn(x, (gb32, 32, 1e-5, SILU), w32)
    assert out is not None and not hasattr(x, "saspa_gn")
    ref = _ref_conv(_ref_gn(x, gamma, beta, 32, 1e-5, True), wf, None, None, None)
    got = out.float().cpu()
    assert (got - ref).abs().max().item() < 1.2e-2 * ref.abs().max().item()
    # the same through a producer that leaves epilogue statistics: conv (stats) -> conv_gn reads them
    w0f, _, w064 = _weights(g, c, c, dev)
    mid = ops.conv(x, w064, None, kh=3, kw=3, pad=1, gn_unit=2)
    assert not hasattr(mid, "saspa_gn")                            # N = 64 cannot carry statistics: still the pass form
    wide = torch.randn(b, h, w_, 320, generator=g).to(dev, BF)
    w1f, w132, w164 = _weights(g, 320, 320, dev)
    prod = ops.conv(wide, w164, None, kh=3, kw=3, pad=1, gn_unit=10)
    assert hasattr(prod, "saspa_gn")
    gamma2 = (1 + 0.2 * torch.randn(320, generator=g)).to(dev)
    beta2 = (0.3 * torch.randn(320, generator=g)).to(dev)
    gb2 = W.pack_gamma_beta32(gamma2.cpu(), beta2.cpu()).to(dev)
    o2 = ops.conv_gn(prod, (gb2, 32, 1e-5, SILU), w132)
    r2 = _ref_conv(_ref_gn(prod, gamma2, beta2, 32, 1e-5, True), w1f, None, None, None)
    assert (o2.float().cpu() - r2).abs().max().item() < 1.2e-2 * r2.abs().max().item()


def test_halo_conv_deferred_reduce_groupnorm(dev):
    """conv1 (norm1 fused, K slices) -> saspa_splitk_groupnorm (reduce + norm2 + SiLU): the chain of a ResnetBlock2D at the
    16x16 level, against groupnorm -> conv(fuse_gn=...)."""
    g = torch.Generator().manual_seed(11)
    b, h, w_, c, n = 2, 16, 16, 256, 320
    x = torch.randn(b, h, w_, c, generator=g).to(dev, BF)
    wf, w32, w64 = _weights(g, n, c, dev)
    bias = torch.randn(n, generator=g).to(dev)
    rv = torch.randn(b, n, generator=g).to(dev)
    g1 = (1 + 0.2 * torch.randn(c, generator=g)).to(dev), (0.3 * torch.randn(c, generator=g)).to(dev)
    g2 = (1 + 0.2 * torch.randn(n, generator=g)).to(dev), (0.3 * torch.randn(n, generator=g)).to(dev)
    gb32 = W.pack_gamma_beta32(g1[0].cpu(), g1[1].cpu()).to(dev)
    got = ops.conv_gn(x, (gb32, 32, 1e-5, SILU), w32, bias, rowvec=rv, ksplit=2, defer_to=(g2[0], g2[1], 32, 1e-5, SILU))
    hn = ops.groupnorm(x, g1[0], g1[1], 32, 1e-5, SILU)
    two = ops.conv(hn, w64, bias, kh=3, kw=3, pad=1, rowvec=rv, ksplit=2, fuse_gn=(g2[0], g2[1], 32, 1e-5, SILU))
    ref = _ref_gn(_ref_conv(_ref_gn(x, g1[0], g1[1], 32, 1e-5, True), wf, bias, rv, None), g2[0], g2[1], 32, 1e-5, True)
    for name, t in (("fused", got), ("two-launch", two)):
        assert (t.float().cpu() - ref).abs().max().item() < 3e-2 * ref.abs().max().item(), name
    assert (got.float() - two.float()).abs().max().item() < 3e-2 * ref.abs().max().item()


def test_halo_conv_ineligible_shapes_return_none(dev):
    x = torch.randn(1, 8, 8, 64).to(dev, BF)                        # 64 pixels per image
    w32 = torch.randn(320, 576).to(dev, BF)
    assert ops.conv_gn(x, None, w32) is None
    x = torch.randn(1, 16, 16, 48).to(dev, BF)                      # channel count not a multiple of 64
    assert ops.conv_gn(x, None, torch.randn(320, 432).to(dev, BF)) is None
    x = torch.randn(1, 16, 16, 64).to(dev, BF)
    assert ops.conv_gn(x, None, torch.randn(96, 576).to(dev, BF)) is None      # N neither a multiple of 320 nor of 256

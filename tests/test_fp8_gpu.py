"""fp8 (OCP e4m3) W8A8 linears of the SDXL transformer blocks (BASELINE.json configs[4]): the quantising
LayerNorm and the 128x128x128 fp8 GEMM against torch (exact on the quantised operands, and the quantisation error itself
against the unquantised layer)."""
import math

import pytest
import torch
import torch.nn.functional as F

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
from saspa_aug_amd import weights as W

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("rows,c", [(300, 640), (77, 1280), (1000, 128), (5, 2048)])
def test_layernorm_quant_fp8(dev, rows, c):
    x = (_rand(rows, c, seed=1) * 2 + 0.5).bfloat16()
    g, b = 1 + 0.1 * _rand(c, seed=2), 0.1 * _rand(c, seed=3)
    q, sc = ops.layernorm_quant_fp8(x.to(dev), g.to(dev), b.to(dev))
    y = F.layer_norm(x.float(), (c,), g, b, 1e-5)
    want_sc = y.abs().amax(1) / 448.0
    assert torch.allclose(sc.cpu(), want_sc, rtol=1e-5, atol=0)
    deq = q.cpu().view(torch.float8_e4m3fn).float() * sc.cpu()[:, None]
    # e4m3: 3 mantissa bits -> half an ulp = 2^-4 relative, plus the subnormal floor of the row scale
    err = (deq - y).abs()
    assert (err <= 0.0625 * y.abs() + sc.cpu()[:, None] * 2 ** -9 + 1e-6).all(), err.max()
    # the same bytes torch produces from the same scaled values (round to nearest even, no saturation needed)
    want_q = (y / sc.cpu()[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    assert (q.cpu() != want_q).float().mean().item() < 1e-3          # last-bit differences of the fp32 LayerNorm only


@pytest.mark.parametrize("m,k,n,geglu,res", [(1000, 640, 640, False, True), (4096, 1280, 1280, False, False),
                                             (515, 640, 5120, True, False), (2048, 1280, 10240, True, False),
                                             (130, 128, 128, False, True)])
def test_gemm_fp8(dev, m, k, n, geglu, res):
    xq = torch.randn(m, k, generator=torch.Generator().manual_seed(4)).clamp(-3, 3)
    sa = 0.5 + torch.rand(m, generator=torch.Generator().manual_seed(5))
    xq8 = (xq * 100).to(torch.float8_e4m3fn)                          # arbitrary representable e4m3 values
    w = _rand(n, k, seed=6, scale=1 / math.sqrt(k))
    b = _rand(n, seed=7)
    if geglu:
        w, b = W.pack_geglu_tile(w, b, 128)
    wq, sw = W.quantize_fp8(w)
    r = _rand(m, n, seed=8).bfloat16() if res else None
    ref = (xq8.double() * sa.double()[:, None]) @ W.dequantize_fp8(wq, sw).double().t() + b.double()
    if geglu:
        # un-pack: tile t = [values of features 64t .. 64t+63 | their gates]
        rt = ref.view(m, n // 128, 2, 64)
        ref = (rt[:, :, 0] * F.gelu(rt[:, :, 1])).reshape(m, n // 2)
    if res:
        ref = ref + r.double()
    out = ops.linear_fp8(xq8.view(torch.uint8).to(dev), sa.to(dev), wq.to(dev), sw.to(dev), b.to(dev),
                         residual=None if r is None else r.to(dev), act=ops.ACT_GEGLU if geglu else ops.ACT_NONE)
    got = out.float().cpu().double()
    err = (got - ref).abs()
    tol = 2 ** -7 * ref.abs() + 2e-2 * ref.abs().max() / 16                       # bf16 output rounding + fp32 accumulation
    assert (err <= tol).all(), (err.max().item(), ref.abs().max().item())


@pytest.mark.parametrize("m,k,n", [(515, 640, 5120), (2048, 1280, 10240)])
def test_gemm_fp8_geglu_emits_fp8_under_a_tensor_scale(dev, m, k, n):
    """ABI 20: the GEGLU epilogue of saspa_gemm_fp8 with `amax` (calibration) and `out_fp8` (e4m3 bytes under ONE power-of-two
    scale).  The bf16-output launch of the same operands is the reference: amax = its largest magnitude (up to the bf16 rounding the
    reference carries), the dequantised bytes are within half an e4m3 ulp of it, a power-of-two change of the scale leaves the
    decoded values unchanged (the format is floating: the exponent shifts), a far-too-small scale saturates at +-448 instead of
    producing NaN bytes."""
    xq8 = (torch.randn(m, k, generator=torch.Generator().manual_seed(4)).clamp(-3, 3) * 100).to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
    sa = (0.5 + torch.rand(m, generator=torch.Generator().manual_seed(5))).to(dev)
    w, b = W.pack_geglu_tile(_rand(n, k, seed=6, scale=1 / math.sqrt(k)), _rand(n, seed=7), 128)
    wq, sw = W.quantize_fp8(w)
    wq, sw, b = wq.to(dev), sw.to(dev), b.to(dev)
    ref = ops.linear_fp8(xq8, sa, wq, sw, b, act=ops.ACT_GEGLU).float()
    amax = torch.zeros(1, device=dev)
    again = ops.linear_fp8(xq8, sa, wq, sw, b, act=ops.ACT_GEGLU, amax=amax).float()
    assert torch.equal(ref, again)                                         # measuring does not change the bf16 result
    a = amax.item()
    assert abs(a - ref.abs().max().item()) <= 2 ** -8 * a
    scale = ops.fp8_pow2_scale(amax)
    sv = scale.item()
    assert math.log2(sv) == round(math.log2(sv)) and 16 * a / 448 <= sv < 32 * a / 448
    q = ops.linear_fp8(xq8, sa, wq, sw, b, act=ops.ACT_GEGLU, out_fp8_scale=scale)
    assert q.dtype == torch.uint8 and q.shape == (m, n // 2)
    deq = q.view(torch.float8_e4m3fn).float() * sv
    err = (deq - ref).abs()
    assert (err <= (0.0625 + 2 ** -8) * ref.abs() + sv * 2 ** -9 + 1e-6).all(), err.max().item()
    q4 = ops.linear_fp8(xq8, sa, wq, sw, b, act=ops.ACT_GEGLU, out_fp8_scale=scale / 4)
    d4 = q4.view(torch.float8_e4m3fn).float() * (sv / 4)
    big = ref.abs() > sv * 2 ** -4                                           # away from either scale's subnormal range
    assert torch.equal(d4[big], deq[big])
    qs = ops.linear_fp8(xq8, sa, wq, sw, b, act=ops.ACT_GEGLU, out_fp8_scale=scale / 4096)
    ds = qs.view(torch.float8_e4m3fn).float()
    assert torch.isfinite(ds).all() and ds.abs().max().item() == 448.0
    # the output projection that reads the bytes: ONE scale for every row == the same scale spelled out per row
    w2q, sw2 = W.quantize_fp8(_rand(640, n // 2, seed=9, scale=1 / math.sqrt(n // 2)))
    r = _rand(m, 640, seed=10).bfloat16().to(dev)
    o1 = ops.linear_fp8(q, scale, w2q.to(dev), sw2.to(dev), residual=r)
    o2 = ops.linear_fp8(q, scale.expand(m).contiguous(), w2q.to(dev), sw2.to(dev), residual=r)
    assert torch.equal(o1, o2)


def test_fp8_gemm_refuses_fp8_output_outside_the_geglu_epilogue(dev):
    x8 = torch.zeros(128, 128, dtype=torch.uint8, device=dev)
    one = torch.ones(1, device=dev)
    with pytest.raises(ValueError):
        ops.linear_fp8(x8, one.expand(128).contiguous(), x8, one.expand(128).contiguous(), out_fp8_scale=one)


def test_fp8_linear_quantisation_error_vs_bf16_layer(dev):
    """LayerNorm -> Linear: W8A8 (per-token x per-channel scales) against the fp64 layer, next to the bf16 path's error."""
    m, c, n = 2048, 1280, 1280
    x = (_rand(m, c, seed=9) * 1.5 + 0.3).bfloat16()
    g, be = 1 + 0.1 * _rand(c, seed=10), 0.1 * _rand(c, seed=11)
    w = _rand(n, c, seed=12, scale=1 / math.sqrt(c))
    ref = F.layer_norm(x.double(), (c,), g.double(), be.double(), 1e-5) @ w.double().t()
    wq, sw = W.quantize_fp8(w)
    q, sc = ops.layernorm_quant_fp8(x.to(dev), g.to(dev), be.to(dev))
    o8 = ops.linear_fp8(q, sc, wq.to(dev), sw.to(dev)).float().cpu().double()
    o16 = ops.linear(ops.layernorm(x.to(dev), g.to(dev), be.to(dev)), w.bfloat16().to(dev)).float().cpu().double()[:, :n]
    rel = lambda y: ((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()     # noqa: E731
    print(f"LN -> linear {m}x{c}x{n}: rms-rel error fp8 {rel(o8):.3e}, bf16 {rel(o16):.3e}")
    assert rel(o8) < 4e-2 and rel(o16) < 6e-3


def test_sdxl_pipeline_fp8_vs_oracle_and_bf16(dev):
    """`enable_fp8()` on a reduced-width SDXL whose transformer widths are multiples of 128 (so the fp8 kernels really run):
    the W8A8 projections keep the 2-step trajectory within the bf16 path's distance class of the oracle, deterministically."""
    import numpy as np
    from oracle import pipeline as OP
    from oracle.canny import generate_canny_array
    from saspa_aug_amd import config as CFG
    from saspa_aug_amd.pipeline import StableDiffusionXLControlNetPipeline
    from saspa_aug_amd.synthetic import synthetic_image
    from tests.util import from_nhwc
    cfgs = CFG.tiny_xl(width=64)                                     # levels 1 / 2 carry the transformers: 128- and 256-wide
    fam = W.synth_family(cfgs, seed=2)
    b, res, steps = 2, 64, 2
    v = cfgs["text"]["vocab"]
    rs = np.random.RandomState(3)
    ids1 = np.full((b, 77), v - 1, np.int64)
    ids1[:, 0] = v - 2
    ids1[:, 1:9] = rs.randint(0, v - 2, (b, 8))
    ctrls = np.stack([generate_canny_array(synthetic_image(res, res, 30 + i), 120, 200) for i in range(b)])
    lat = torch.randn((b, 4, res // 8, res // 8), generator=torch.manual_seed(1), dtype=torch.float16)
    outs = {}
    for mode in ("bf16", "fp8"):
        pipe = StableDiffusionXLControlNetPipeline(dict(fam), cfgs)
        if mode == "fp8":
            pipe.enable_fp8()
        pipe = pipe.to(dev, torch.bfloat16)
        if mode == "fp8":
            assert len(pipe.unet.fp8_blocks) > 0 and len(pipe.controlnet.fp8_blocks) > 0
            # round 6 breadth: the fused self-attention projection and the feed-forward output projection too
            assert len(pipe.unet.fp8_qkv) == len(pipe.unet.fp8_blocks) and len(pipe.unet.fp8_ffout) == len(pipe.unet.fp8_blocks)
            assert not any(pipe.unet.fp8_ffout.values())                       # scales not calibrated before the first evaluation
        ids2 = pipe.pad_ids_2(ids1)
        out, x, img = pipe.generate_batch(ids1, None, ctrls, lat, steps, 0.0, 0.75, return_latents=True, prompt_ids_2=ids2)
        out2, _, _ = pipe.generate_batch(ids1, None, ctrls, lat, steps, 0.0, 0.75, return_latents=True, prompt_ids_2=ids2)
        assert torch.equal(out, out2)
        if mode == "fp8":
            assert all(pipe.unet.fp8_ffout.values()) and all(pipe.controlnet.fp8_ffout.values())
            sc = [pipe.unet.p[t + ".ff.scale"].item() for t in pipe.unet.fp8_ffout]
            assert all(v > 0 and math.log2(v) == round(math.log2(v)) for v in sc)    # calibrated powers of two
        outs[mode] = from_nhwc(x, 4)
    refs = [OP.sdxl_controlnet_pipeline(fam, cfgs, torch.from_numpy(ids1[i:i + 1]), torch.from_numpy(ids2[i:i + 1]), ctrls[i],
                                        lat[i:i + 1].float(), steps, return_latents=True)[1] for i in range(b)]
    ref = torch.cat(refs)
    rel = lambda y: ((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()     # noqa: E731
    e16, e8 = rel(outs["bf16"]), rel(outs["fp8"])
    print(f"SDXL (reduced width) final latents vs oracle, rms-rel: bf16 {e16:.3e}, fp8 projections {e8:.3e}")
    assert e16 < 3e-2 and e8 < 8e-2, (e16, e8)

"""Parity of the PRODUCTION path at the benchmark configuration (BASELINE.json configs[1]): SD-v1.5 + Canny ControlNet at
full width, batch 8, 512x512, bf16 MFMA kernels, hipGraph replay of the sampling step -- the path bench.py times
(reference call sites run_aug/run_aug.py:235-241, :268-269, :278).

What is compared with what:
  (a) bf16 + hipGraph, batch 8, against the CPU oracle on the same weights / token ids / control image / noise for a short
      DDIM trajectory of images 0 and 7 (the oracle costs ~6 s per CFG evaluation at 512x512, so 5 steps each): final
      latents, decoded image, and -- teacher-forced on the oracle's own latents -- the relative error of every single UNet+ControlNet
      evaluation (no compounding);
  (b) bf16 + hipGraph, batch 8, 50 DDIM steps against this repo's exact-fp32 MFMA path (itself within 1e-5 of the oracle,
      test_models_gpu.py) on the same inputs: GPU only, the whole benchmark trajectory;
  (c) hipGraph replay == launching every kernel from Python, bit for bit, at full width and batch 8;
  (d) the batch-8 result against the same items run one at a time.  Dispatch depends on M (wide 256x320 tiles and K slices
      at batch 8, 128x160 tiles at batch 1; the flash-attention key tile differs too), so the two runs round differently in
      every bf16 tensor: NOT bit-exact (measured, printed).  What is asserted: the two bf16 realisations are no further
      from each other than 2x the larger of their own distances from the exact-fp32 path -- batching changes an image by
      bf16 rounding noise only.

Bounds are 2x the values measured on the MI355X (printed by the tests; DESIGN.md section 5 quotes them)."""
import numpy as np
import pytest
import torch

import saspa_aug_amd  # noqa: F401
from oracle import pipeline as OP
from oracle.canny import generate_canny_array
from saspa_aug_amd import config as CFG
from saspa_aug_amd import ops
from saspa_aug_amd import weights as W
from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline, graphs_enabled
from saspa_aug_amd.synthetic import negative_prompt_ids, synthetic_image, synthetic_prompt_ids
from tests.util import from_nhwc, to_nhwc

pytestmark = pytest.mark.gpu

B, RES = 8, 512


def _psnr(a, b):
    mse = float((a.double() - b.double()).pow(2).mean())
    return 10 * np.log10(1.0 / max(mse, 1e-20))


def _img01(img_nhwc_or_nchw, nchw=False):
    x = img_nhwc_or_nchw if nchw else from_nhwc(img_nhwc_or_nchw, 3)
    return (x / 2 + 0.5).clamp(0, 1)


@pytest.fixture(scope="module")
def prod(dev):
    cfgs = {k: v for k, v in CFG.SD15.items() if k != "safety"}       # the checker has its own parity tests
    fam = W.synth_family(cfgs, seed=0)
    pipe = StableDiffusionControlNetPipeline(dict(fam), cfgs).to(dev, torch.bfloat16)
    vocab = cfgs["text"]["vocab"]
    ids = synthetic_prompt_ids(B, seed=1, vocab=vocab)
    neg = negative_prompt_ids(vocab)
    ctrls = np.stack([generate_canny_array(synthetic_image(RES, RES, 40 + i), 120, 200) for i in range(B)])
    g = torch.Generator().manual_seed(1)
    lat = torch.randn((B, 4, RES // 8, RES // 8), generator=g, dtype=torch.float16)      # the reference's pipeline dtype
    state = dict(cfgs=cfgs, fam=fam, pipe=pipe, ids=ids, neg=neg, ctrls=ctrls, lat=lat)
    yield state
    state.clear()
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def p32(dev, prod):
    """The exact-fp32 MFMA path on the same weights (within 1e-5 of the oracle: test_models_gpu.py)."""
    pipe = StableDiffusionControlNetPipeline(dict(prod["fam"]), prod["cfgs"]).to(dev, torch.float32)
    yield pipe
    del pipe
    torch.cuda.empty_cache()


def _rms_rel(a, b):
    return ((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt()).item()


def test_bf16_graph_batch8_vs_oracle_short_trajectory(dev, prod):
    """(a) 5 DDIM steps, batch 8 through the bf16 + hipGraph path; images 0 AND 7 of the batch against the oracle (two
    positions, different prompts / control images / noise, so an error that depends on the batch index cannot hide behind
    image 0)."""
    cfgs, fam, pipe = prod["cfgs"], prod["fam"], prod["pipe"]
    steps = 5
    assert graphs_enabled()
    out, x, img = pipe.generate_batch(prod["ids"], prod["neg"], prod["ctrls"], prod["lat"], steps, return_latents=True)
    assert any(g.graph is not None for g in pipe._graphs.values()), "the hipGraph replay path did not run"
    out, x, img = out.clone(), x.clone(), img.clone()
    res, trace0 = {}, None
    for i in (0, 7):
        trace = []
        ref_u8, ref_x, ref_img = OP.sd_controlnet_pipeline(
            fam, cfgs, torch.from_numpy(prod["ids"][i:i + 1]), torch.from_numpy(prod["neg"]), prod["ctrls"][i],
            prod["lat"][i:i + 1].float(), steps, return_latents=True, trace=trace)
        if i == 0:
            trace0 = trace
        got_x = from_nhwc(x[i:i + 1], 4)
        got01, ref01 = _img01(img[i:i + 1]), _img01(ref_img, nchw=True)
        du8 = np.abs(out[i:i + 1].cpu().numpy().astype(int) - ref_u8.astype(int))
        res[i] = dict(lat_rel=((got_x - ref_x).abs().max() / ref_x.abs().max()).item(),
                      lat_rms=((got_x - ref_x).pow(2).mean().sqrt() / ref_x.pow(2).mean().sqrt()).item(),
                      d01=(got01 - ref01).abs().max().item(), psnr=_psnr(got01, ref01), du8_max=int(du8.max()),
                      du8_mean=float(du8.mean()))
    # cross check: image 7 of the batch must NOT look like image 0's reference (the two items are different work)
    cross = _psnr(_img01(img[7:8]), _img01(img[0:1]))
    # teacher forcing: every oracle step's latents through the bf16 networks (eager launches), eps against the oracle's
    unet, cn = pipe.unet, pipe.controlnet
    trace = trace0
    ts = [tr["t"] for tr in trace]
    ctx = trace[0]["ctx"].to(dev, torch.bfloat16)
    for net in (unet, cn):
        net.prepare_context(ctx)
        net.prepare_timesteps(ts)
    cond = OP.prepare_control(prod["ctrls"][0])
    cemb = cn.cond_embedding(to_nhwc(torch.cat([cond, cond]), torch.bfloat16, dev, cpad=8))
    eps_rel, eps_rms = [], []
    for k, tr in enumerate(trace):
        xd = to_nhwc(torch.cat([tr["x"], tr["x"]]), torch.bfloat16, dev, cpad=8)
        mid, skips = unet.encode(xd, k)
        s2, m2 = cn.forward(xd, k, cemb, 0.75, skips, mid)
        e = from_nhwc(unet.decode(m2, s2, k), 4)
        eps_rel.append(((e - tr["eps2"]).abs().max() / tr["eps2"].abs().max()).item())
        eps_rms.append(((e - tr["eps2"]).pow(2).mean().sqrt() / tr["eps2"].pow(2).mean().sqrt()).item())
    for i, r in res.items():
        print(f"\n[production a] bf16+graph batch 8 vs oracle, {steps} steps, image {i}: latents max-rel {r['lat_rel']:.3e} rms-rel "
              f"{r['lat_rms']:.3e}; image max|d| {r['d01']:.4f} PSNR {r['psnr']:.1f} dB, u8 max {r['du8_max']} mean {r['du8_mean']:.3f}")
    print(f"[production a] image 7 vs image 0 of the same batch: PSNR {cross:.1f} dB (different items); "
          f"per-evaluation eps max-rel {max(eps_rel):.3e} rms-rel {max(eps_rms):.3e} ({['%.2e' % v for v in eps_rms]})")
    # measured on MI355X (round 2, current synthetic weights): eps rms-rel 1.35e-2 at t=981, 8.3e-3..1.0e-2 after; max-rel
    # 1.31e-2; latents rms-rel 2.66e-2 (the CFG combine amplifies uncorrelated eps error by ~7.5*sqrt(2)); image PSNR 41.4 dB.
    # Bounds = 2x (PSNR - 6 dB).
    assert max(eps_rms) < 2.7e-2 and max(eps_rel) < 3.0e-2, (eps_rel, eps_rms)
    for i, r in res.items():
        assert r["lat_rms"] < 5.4e-2 and r["psnr"] > 35.4, (i, r)
    assert cross < 30.0, cross


def test_bf16_graph_batch8_nonsquare_vs_oracle(dev, prod):
    """The dominant bucket of BASELINE configs[3] (FGVC-Aircraft resized by all_utils/utils.py:58-79 -> 512x704): batch 8,
    bf16 + hipGraph, 3 DDIM steps, image 0 against the CPU oracle -- the same comparison as (a) at the size the dispatch
    heuristics were NOT tuned on (M = 90 112 / 22 528 / 5 632 / 1 408 rows, 5 632-token attention)."""
    cfgs, fam, pipe = prod["cfgs"], prod["fam"], prod["pipe"]
    hh, ww, steps = 512, 704, 3
    ctrls = np.stack([generate_canny_array(synthetic_image(hh, ww, 140 + i), 120, 200) for i in range(B)])
    lat = torch.randn((B, 4, hh // 8, ww // 8), generator=torch.Generator().manual_seed(2), dtype=torch.float16)
    out, x, img = pipe.generate_batch(prod["ids"], prod["neg"], ctrls, lat, steps, return_latents=True)
    assert torch.isfinite(from_nhwc(img, 3)).all()
    ref_u8, ref_x, ref_img = OP.sd_controlnet_pipeline(fam, cfgs, torch.from_numpy(prod["ids"][:1]), torch.from_numpy(prod["neg"]),
                                                       ctrls[0], lat[:1].float(), steps, return_latents=True)
    got_x = from_nhwc(x[:1], 4)
    lat_rms = ((got_x - ref_x).pow(2).mean().sqrt() / ref_x.pow(2).mean().sqrt()).item()
    got01, ref01 = _img01(img[:1]), _img01(ref_img, nchw=True)
    d01, psnr = (got01 - ref01).abs().max().item(), _psnr(got01, ref01)
    print(f"\n[production 512x704] bf16+graph batch 8 vs oracle, {steps} steps, image 0: latents rms-rel {lat_rms:.3e}; "
          f"image max|d| {d01:.4f} PSNR {psnr:.1f} dB")
    # same bounds as (a) (2x the 512x512 measurements): the non-square bucket must be no worse than the tuned size
    assert lat_rms < 5.4e-2 and psnr > 35.4, (lat_rms, d01, psnr)


def test_bf16_vs_fp32_hip_batch8_50_steps(dev, prod, p32):
    """(b) the benchmark trajectory (batch 8, 50 DDIM steps, bf16, graph) against the exact-fp32 HIP path, item by item."""
    cfgs, fam, pipe = prod["cfgs"], prod["fam"], prod["pipe"]
    steps = 50
    out, x, img = pipe.generate_batch(prod["ids"], prod["neg"], prod["ctrls"], prod["lat"], steps, return_latents=True)
    got01 = _img01(img)
    assert torch.isfinite(got01).all()
    psnrs, dmax, lat_rms = [], [], []
    for i in (0, 7):              # fp32 runs one image at a time (its unfused attention scores are 1 GiB per image)
        o32, x32, i32 = p32.generate_batch(prod["ids"][i:i + 1], prod["neg"], prod["ctrls"][i:i + 1], prod["lat"][i:i + 1].float(),
                                           steps, return_latents=True)
        r01 = _img01(i32)
        psnrs.append(_psnr(got01[i:i + 1], r01))
        dmax.append((got01[i:i + 1] - r01).abs().max().item())
        a, b = from_nhwc(x[i:i + 1], 4), from_nhwc(x32, 4)
        lat_rms.append(((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item())
    print(f"\n[production b] bf16+graph batch 8 vs fp32 HIP, 50 steps: PSNR {['%.1f' % v for v in psnrs]} dB, max|d| "
          f"{['%.3f' % v for v in dmax]}, latents rms-rel {['%.2e' % v for v in lat_rms]}")
    # measured (round 2): PSNR 44.2-44.6 dB (43.8-45.2 with the first weight set), latents rms-rel 1.7e-2..1.8e-2.
    # Bounds = 2x the error (PSNR - 6 dB).
    assert min(psnrs) > 37.8 and max(lat_rms) < 4.0e-2, (psnrs, dmax, lat_rms)


def test_graph_replay_equals_eager_full_width_batch8(dev, prod, monkeypatch):
    """(c) bit-exact: the captured step replays the same kernels in the same order."""
    pipe = prod["pipe"]
    steps = 3
    a_out, a_x, _ = pipe.generate_batch(prod["ids"], prod["neg"], prod["ctrls"], prod["lat"], steps, return_latents=True)
    a_out, a_x = a_out.clone(), a_x.clone()
    monkeypatch.setenv("SASPA_GRAPH", "0")
    assert not graphs_enabled()
    b_out, b_x, _ = pipe.generate_batch(prod["ids"], prod["neg"], prod["ctrls"], prod["lat"], steps, return_latents=True)
    assert torch.equal(a_x, b_x), "hipGraph replay differs from the eager launch loop at full width"
    assert torch.equal(a_out, b_out)


def test_batch8_vs_single_items_full_width(dev, prod, p32):
    """(d) batch 8 against the same items run alone (3 steps), both measured against the exact-fp32 path."""
    pipe = prod["pipe"]
    steps = 3
    out, x, img = pipe.generate_batch(prod["ids"], prod["neg"], prod["ctrls"], prod["lat"], steps, return_latents=True)
    out, x = out.clone(), x.clone()
    exact, diff, e8, e1, du8 = [], [], [], [], []
    for i in (0, 5):
        sl = slice(i, i + 1)
        o1, x1, _ = pipe.generate_batch(prod["ids"][sl], prod["neg"], prod["ctrls"][sl], prod["lat"][sl], steps, return_latents=True)
        x1, o1 = x1.clone(), o1.clone()
        _, x32, _ = p32.generate_batch(prod["ids"][sl], prod["neg"], prod["ctrls"][sl], prod["lat"][sl].float(), steps,
                                       return_latents=True)
        exact.append(bool(torch.equal(x1, x[sl])))
        diff.append(_rms_rel(x1.float(), x[sl].float()))
        e8.append(_rms_rel(x[sl].float(), x32))
        e1.append(_rms_rel(x1.float(), x32))
        du8.append(int((o1.int() - out[sl].int()).abs().max()))
    print(f"\n[production d] batch 8 vs single items, 3 steps: bit-exact {exact}; latents rms-rel between the two "
          f"{['%.2e' % v for v in diff]}; batch-8 vs fp32 {['%.2e' % v for v in e8]}; single vs fp32 {['%.2e' % v for v in e1]}; "
          f"u8 max diff {du8}")
    for d, a8, a1 in zip(diff, e8, e1):
        assert d < 2.0 * max(a8, a1), (diff, e8, e1)
    assert max(e8) < 6e-2 and max(e1) < 6e-2, (e8, e1)

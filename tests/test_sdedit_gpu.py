"""SDEdit / img2img (run_aug/run_aug.py:203-206, :252-260, :274-276): the VAE encoder launch graph, the fused
posterior-sample + add-noise kernel and the whole StableDiffusionControlNetImg2ImgPipeline against the oracle (fp32 path,
north-star bar), the reference's call form, and the batched run_aug path with SDEDIT = 1."""
import numpy as np
import pytest
import torch
from PIL import Image

import saspa_aug_amd  # noqa: F401
from oracle import pipeline as OP
from oracle import sd_models as OM
from oracle.canny import generate_canny_array
from saspa_aug_amd import config as CFG
from saspa_aug_amd import models, ops
from saspa_aug_amd import run_aug as R
from saspa_aug_amd import weights as W
from saspa_aug_amd.pipeline import StableDiffusionControlNetImg2ImgPipeline, StableDiffusionImg2ImgPipeline
from saspa_aug_amd.synthetic import synthetic_image
from tests.util import from_nhwc, to_nhwc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    cfgs = {k: v for k, v in CFG.tiny().items() if k != "safety"}
    return cfgs, W.synth_family(cfgs, seed=3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vae_encoder(dev, tiny, dtype):
    cfgs, fam = tiny
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(1)) * 2 - 1
    mean, logvar = OM.vae_encode(fam["vae"], cfgs["vae"], x)
    got = from_nhwc(models.VAEEncoder(fam["vae"], cfgs["vae"], dev, dtype).encode(to_nhwc(x, dtype, dev, cpad=8)))
    ref = torch.cat([mean, logvar], 1)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    assert got.shape == ref.shape and err < (2e-4 if dtype == torch.float32 else 6e-2), err


def test_vae_encoder_full_width(dev):
    cfg = CFG.SD15["vae"]
    sd = W.synth_state_dict("vae_enc", cfg, 5)
    x = torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(2)) * 2 - 1
    mean, logvar = OM.vae_encode(sd, cfg, x)
    got = from_nhwc(models.VAEEncoder(sd, cfg, dev, torch.float32).encode(to_nhwc(x, torch.float32, dev, cpad=8)))
    ref = torch.cat([mean, logvar], 1)
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 3e-4


def test_sample_noise_kernel(dev):
    g = torch.Generator().manual_seed(3)
    mom = torch.randn(2, 8, 8, 8, generator=g)
    mom[0, 0, 0, 4] = 50.0           # clamp(logvar, -30, 20)
    mom[0, 0, 1, 5] = -60.0
    e1, e2 = torch.randn(2, 8, 8, 8, generator=g), torch.randn(2, 8, 8, 8, generator=g)
    got = ops.vae_sample_noise(mom.to(dev), e1.to(dev), e2.to(dev), 0.18215, 0.6, 0.8).cpu()
    x0 = (mom[..., :4] + torch.exp(0.5 * mom[..., 4:].clamp(-30, 20)) * e1[..., :4]) * 0.18215
    ref = 0.6 * x0 + 0.8 * e2[..., :4]
    assert torch.allclose(got[..., :4], ref, rtol=1e-5, atol=1e-5) and (got[..., 4:] == 0).all()


def _case(cfgs, fam, dev, dtype, steps, strength, hh=64, ww=64, nimg=2):
    ids = torch.from_numpy(np.random.RandomState(1).randint(0, cfgs["text"]["vocab"] - 2, (nimg, 77)))
    neg = torch.from_numpy(np.random.RandomState(2).randint(0, cfgs["text"]["vocab"] - 2, (1, 77)))
    srcs = np.stack([synthetic_image(hh, ww, 20 + i) for i in range(nimg)])
    ctrls = np.stack([generate_canny_array(s, 120, 200) for s in srcs])
    g = torch.manual_seed(1)
    e = [torch.randn((1, 4, hh // 8, ww // 8), generator=g) for _ in range(2 * nimg)]
    e1, e2 = torch.cat(e[0::2]), torch.cat(e[1::2])
    refs = [OP.sd_controlnet_img2img_pipeline(fam, cfgs, ids[i:i + 1], neg, srcs[i], ctrls[i], e1[i:i + 1], e2[i:i + 1], steps, strength,
                                              return_latents=True) for i in range(nimg)]
    pipe = StableDiffusionControlNetImg2ImgPipeline(fam, cfgs).to(dev, dtype)
    out, x, img = pipe.generate_batch_img2img(ids.numpy(), neg.numpy(), srcs, ctrls, e1, e2, steps, strength, return_latents=True)
    ref_img = torch.cat([r[2] for r in refs])
    d01 = ((from_nhwc(img, 3) / 2 + 0.5).clamp(0, 1) - (ref_img / 2 + 0.5).clamp(0, 1)).abs().max().item()
    du8 = int(np.abs(out.cpu().numpy().astype(int) - np.concatenate([r[0] for r in refs]).astype(int)).max())
    return d01, du8, pipe


@pytest.mark.parametrize("graph", ["1", "0"])
def test_img2img_pipeline_fp32_parity(dev, tiny, graph, monkeypatch):
    """SDEDIT_STRENGTH = 0.85 of 10 steps -> the last 8 run (get_timesteps); strength 0.5 -> 5; atol 1e-3 per pixel."""
    monkeypatch.setenv("SASPA_GRAPH", graph)
    cfgs, fam = tiny
    assert StableDiffusionControlNetImg2ImgPipeline.kept_steps(10, 0.85) == (2, 8)
    assert StableDiffusionControlNetImg2ImgPipeline.kept_steps(30, 0.85) == (5, 25)
    for steps, strength in (((10, 0.85), (10, 0.5)) if graph == "1" else ((10, 0.85),)):
        d01, du8, _ = _case(cfgs, fam, dev, torch.float32, steps, strength)
        assert d01 < 1e-3 and du8 <= 1, (steps, strength, d01, du8)


def test_img2img_call_form_and_run_aug(dev, tiny, tmp_path):
    cfgs, fam = tiny
    pipe = R.init_pipeline("sd_v1.5", "canny", 1, cfgs=cfgs, state_dicts=fam)
    assert isinstance(pipe, StableDiffusionControlNetImg2ImgPipeline)
    pipe = pipe.to("cuda:0", torch.float16)
    src = Image.fromarray(synthetic_image(64, 128, 4))
    ctrl = Image.fromarray(generate_canny_array(np.asarray(src), 120, 200))
    a = R.pass_thorugh_pipe("sd_v1.5", pipe, "an airplane", src, 1, 0.85, 6, torch.manual_seed(1), 7.5, 0.75, control_image=ctrl)
    b = pipe(prompt="an airplane", image=src, control_image=ctrl, strength=0.85, num_inference_steps=6, generator=torch.manual_seed(1),
             guidance_scale=7.5, negative_prompt=R.NEGATIVE_PROMPT, controlnet_conditioning_scale=0.75).images[0]
    assert a.size == (128, 64) and np.array_equal(np.asarray(a), np.asarray(b))
    with pytest.raises(ValueError):
        pipe(prompt="x", image=src, control_image=ctrl, strength=0.05, num_inference_steps=6, generator=torch.manual_seed(1))
    # batched loop with SDEDIT = 1: files + JSON, output folder carries the strength (run_aug/run_aug.py:678-680)
    prompts = tmp_path / "p.txt"
    prompts.write_text("A white airplane on a runway.\\nAn airplane above the clouds.\\n")
    s = R.Settings(DATASET="synthetic", NUM_PER_IMAGE=2, SEED=1, RESOLUTION=64, BATCH_SIZE=3, PROMPTS_FILE=str(prompts), SDEDIT=1,
                   NUM_INFERENCE_STEPS=4, SEMANTIC_FILTERING=0, MODEL_CONFIDENCE_BASED_FILTERING=0,
                   DATASET_KWARGS=dict(root_path=str(tmp_path / "ds/data"), n_images=3, sizes=((64, 64),), seed=3))
    res = R.main(s, pipe=pipe)
    assert (res["status"] == 1).all() and "sd_v1.5-SDEdit_strength_0.85" in res["output_folder"]
    # item 0 of the batched run == the single call with the same two draws from the seed-1 stream
    it = res["items"][0]
    g = torch.manual_seed(1)
    single = pipe(prompt=it.prompt, image=Image.open(it.source_path).convert("RGB"),
                  control_image=Image.fromarray(generate_canny_array(np.array(Image.open(it.source_path).convert("RGB")), 120, 200)),
                  strength=0.85, num_inference_steps=4, generator=g, guidance_scale=7.5, negative_prompt=R.NEGATIVE_PROMPT,
                  controlnet_conditioning_scale=0.75).images[0]
    assert np.array_equal(np.asarray(single), np.array(Image.open(it.output_path)))


# ---- ControlNet-free img2img: the Real-Guidance form (CONTROLNET = None, SDEDIT = 1; run_aug/run_aug.py:163-165) ----
@pytest.mark.parametrize("graph", ["1", "0"])
def test_plain_img2img_pipeline_fp32_parity(dev, tiny, graph, monkeypatch):
    """StableDiffusionImg2ImgPipeline against its oracle restatement: strength 0.15 of 50 steps keeps 7 (the Real-Guidance
    operating point, run_aug/run_aug_real_guidance.py:520-523), 0.5 of 10 keeps 5; atol 1e-3 per pixel, <= 1 u8 level."""
    monkeypatch.setenv("SASPA_GRAPH", graph)
    cfgs, fam = tiny
    assert StableDiffusionImg2ImgPipeline.kept_steps(50, 0.15) == (43, 7)
    nimg, hh, ww = 2, 64, 128
    ids = torch.from_numpy(np.random.RandomState(1).randint(0, cfgs["text"]["vocab"] - 2, (nimg, 77)))
    neg = torch.from_numpy(np.random.RandomState(2).randint(0, cfgs["text"]["vocab"] - 2, (1, 77)))
    srcs = np.stack([synthetic_image(hh, ww, 30 + i) for i in range(nimg)])
    g = torch.manual_seed(1)
    e = [torch.randn((1, 4, hh // 8, ww // 8), generator=g) for _ in range(2 * nimg)]
    e1, e2 = torch.cat(e[0::2]), torch.cat(e[1::2])
    pipe = StableDiffusionImg2ImgPipeline(fam, cfgs).to(dev, torch.float32)
    assert pipe.controlnet is None
    for steps, strength in (((50, 0.15), (10, 0.5)) if graph == "1" else ((10, 0.5),)):
        refs = [OP.sd_img2img_pipeline(fam, cfgs, ids[i:i + 1], neg, srcs[i], e1[i:i + 1], e2[i:i + 1], steps, strength, return_latents=True)
                for i in range(nimg)]
        out, x, img = pipe.generate_batch_img2img(ids.numpy(), neg.numpy(), srcs, e1, e2, steps, strength, return_latents=True)
        ref_img = torch.cat([r[2] for r in refs])
        d01 = ((from_nhwc(img, 3) / 2 + 0.5).clamp(0, 1) - (ref_img / 2 + 0.5).clamp(0, 1)).abs().max().item()
        du8 = int(np.abs(out.cpu().numpy().astype(int) - np.concatenate([r[0] for r in refs]).astype(int)).max())
        assert d01 < 1e-3 and du8 <= 1, (steps, strength, d01, du8)
    with pytest.raises(ValueError):          # a control image is refused by the ControlNet-free pipeline
        StableDiffusionControlNetImg2ImgPipeline.generate_batch_img2img(pipe, ids.numpy(), neg.numpy(), srcs, srcs, e1, e2, 10, 0.5)


def test_plain_img2img_call_form_and_run_aug(dev, tiny, tmp_path):
    """init_pipeline("sd_v1.5", None, 1) -> StableDiffusionImg2ImgPipeline; pass_thorugh_pipe's kwargs (:235-241, :274-276);
    the batched loop writes under regular/sd_v1.5-SDEdit_strength_s/None/ (:668-692) and no *_control.png (:435-442)."""
    cfgs, fam = tiny
    pipe = R.init_pipeline("sd_v1.5", None, 1, cfgs=cfgs, state_dicts=fam)
    assert isinstance(pipe, StableDiffusionImg2ImgPipeline)
    pipe = pipe.to("cuda:0", torch.float16)
    src = Image.fromarray(synthetic_image(64, 128, 4))
    a = R.pass_thorugh_pipe("sd_v1.5", pipe, "an airplane", src, 1, 0.5, 6, torch.manual_seed(1), 7.5, 0.75, control_image=None)
    b = pipe(prompt="an airplane", image=src, strength=0.5, num_inference_steps=6, generator=torch.manual_seed(1), guidance_scale=7.5,
             negative_prompt=R.NEGATIVE_PROMPT).images[0]
    assert a.size == (128, 64) and np.array_equal(np.asarray(a), np.asarray(b))
    prompts = tmp_path / "p.txt"
    prompts.write_text("A white airplane on a runway.\nAn airplane above the clouds.\n")
    s = R.Settings(DATASET="synthetic", NUM_PER_IMAGE=2, SEED=1, RESOLUTION=64, BATCH_SIZE=3, PROMPTS_FILE=str(prompts), SDEDIT=1,
                   CONTROLNET=None, SDEDIT_STRENGTH=0.5, NUM_INFERENCE_STEPS=4, SEMANTIC_FILTERING=0, MODEL_CONFIDENCE_BASED_FILTERING=0,
                   DATASET_KWARGS=dict(root_path=str(tmp_path / "ds/data"), n_images=3, sizes=((64, 64),), seed=3))
    res = R.main(s, pipe=pipe)
    assert (res["status"] == 1).all()
    assert "/aug_data/regular/sd_v1.5-SDEdit_strength_0.5/None/" in res["output_folder"]
    import pathlib
    names = [p.name for p in pathlib.Path(res["output_folder"]).glob("*.png")]
    assert not any(n.endswith("_control.png") for n in names) and sum(n.endswith("_source.png") for n in names) == 3
    it = res["items"][0]
    single = pipe(prompt=it.prompt, image=Image.open(it.source_path).convert("RGB"), strength=0.5, num_inference_steps=4,
                  generator=torch.manual_seed(1), guidance_scale=7.5, negative_prompt=R.NEGATIVE_PROMPT).images[0]
    assert np.array_equal(np.asarray(single), np.array(Image.open(it.output_path)))

"""GPU parity of the SDXL-Turbo branch (run_aug/run_aug.py:189-201, :223-228, :564-571) against the CPU
oracle on identical seeded weights / token ids / control image / noise: two text towers read at hidden_states[-2],
pooled text_time conditioning, three-level UNet / ControlNet with deep transformers and linear projections, DDIM
"trailing" without classifier-free guidance, fp32-upcast VAE.

Tolerances as in test_models_gpu.py: fp32 path atol 1e-3 on the decoded image in [0,1] and <= 1 u8 level."""
import numpy as np
import pytest
import torch

import saspa_aug_amd  # noqa: F401
from oracle import pipeline as OP
from oracle import sd_models as OM
from saspa_aug_amd import config as CFG
from saspa_aug_amd import models, ops
from saspa_aug_amd import weights as W
from saspa_aug_amd.pipeline import StableDiffusionXLControlNetPipeline
from saspa_aug_amd.synthetic import synthetic_image
from tests.util import from_nhwc, to_nhwc

pytestmark = pytest.mark.gpu


def _relerr(got, ref):
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()


def _limits(dtype):
    return 3e-4 if dtype == torch.float32 else 8e-2


@pytest.fixture(scope="module")
def tiny_xl():
    cfgs = CFG.tiny_xl()
    return cfgs, W.synth_family(cfgs, seed=3)


def _ids(cfgs, n, seed=1):
    """Token rows as the two tokenizers produce them: BOS, words, EOS then EOS padding (tower 1) / 0 padding (tower 2)."""
    v = cfgs["text"]["vocab"]
    rs = np.random.RandomState(seed)
    ids1 = np.full((n, 77), v - 1, np.int64)
    ids1[:, 0] = v - 2
    for r in range(n):
        k = rs.randint(5, 40)
        ids1[r, 1:1 + k] = rs.randint(0, v - 2, k)
    ids2 = ids1.copy()
    for r in range(n):
        first = int((ids1[r] == v - 1).argmax())
        ids2[r, first + 1:] = 0
    return ids1, ids2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_text_towers_penultimate_and_pooled(dev, tiny_xl, dtype):
    cfgs, fam = tiny_xl
    ids1, ids2 = _ids(cfgs, 3)
    r1, none = OM.clip_text_forward(fam["text"], cfgs["text"], torch.from_numpy(ids1), penultimate=True)
    r2, rp = OM.clip_text_forward(fam["text2"], cfgs["text2"], torch.from_numpy(ids2), penultimate=True)
    assert none is None
    t1 = models.CLIPText(fam["text"], cfgs["text"], dev, dtype)
    t2 = models.CLIPText(fam["text2"], cfgs["text2"], dev, dtype)
    g1, gn = t1.forward(torch.from_numpy(ids1).to(dev), penultimate=True)
    g2, gp = t2.forward(torch.from_numpy(ids2).to(dev), penultimate=True)
    assert gn is None
    lim = _limits(dtype)
    assert _relerr(g1.float().cpu(), r1) < lim and _relerr(g2.float().cpu(), r2) < lim
    assert _relerr(gp.float().cpu()[:, :rp.shape[1]], rp) < lim


def test_pad_ids_2_matches_tokenizer_2_padding(tiny_xl):
    cfgs, fam = tiny_xl
    ids1, ids2 = _ids(cfgs, 4, seed=7)
    pipe = StableDiffusionXLControlNetPipeline(fam, cfgs)
    assert np.array_equal(pipe.pad_ids_2(ids1), ids2)


def _unet_cn_case(cfgs, fam, dev, dtype, b, h, w, cache=None):
    """cache: a dict shared by the dtype variants of one (family, shape) -- the CPU oracle evaluation is computed once."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(b, 4, h, w, generator=g)
    ctx = torch.randn(b, 77, cfgs["unet"]["ctx_dim"], generator=g)
    pooled = torch.randn(b, cfgs["unet"]["add_embed"]["pooled_dim"], generator=g)
    cond = torch.rand(b, 3, 8 * h, 8 * w, generator=g)
    tids = [[8 * h, 8 * w, 0, 0, 8 * h, 8 * w]] * b
    added = dict(text_embeds=pooled, time_ids=torch.tensor(tids, dtype=torch.float32))
    ts = OP.DDIM(spacing="trailing").set_timesteps(2)
    step = 1
    t = int(ts[step])
    if cache is not None and "ref" in cache:
        down, mid, ref = cache["down"], cache["mid"], cache["ref"]
    else:
        down, mid = OM.controlnet_forward(fam["controlnet"], cfgs["controlnet"], x, t, ctx, cond, 0.75, added)
        ref = OM.unet_forward(fam["unet"], cfgs["unet"], x, t, ctx, down, mid, added)
        if cache is not None:
            cache.update(down=down, mid=mid, ref=ref)
    unet = models.UNet(fam["unet"], cfgs["unet"], dev, dtype)
    cn = models.ControlNet(fam["controlnet"], cfgs["controlnet"], dev, dtype)
    assert cn.n_skips == len(down)
    ctxd = ctx.to(dev, dtype)
    for net in (unet, cn):
        net.prepare_context(ctxd)
        net.prepare_timesteps(ts, (pooled.to(dev), tids))
    xd = to_nhwc(x, dtype, dev, cpad=8)
    cemb = cn.cond_embedding(to_nhwc(cond, dtype, dev, cpad=8))
    outs, m = cn.forward(xd, step, cemb, 0.75)
    errs = [_relerr(from_nhwc(o), r) for o, r in zip(outs, down)] + [_relerr(from_nhwc(m), mid)]
    umid, uskips = unet.encode(xd, step)
    s2, m2 = cn.forward(xd, step, cemb, 0.75, uskips, umid)
    got = from_nhwc(unet.decode(m2, s2, step), 4)
    return max(errs), _relerr(got, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unet_controlnet_tiny_xl(dev, tiny_xl, dtype):
    """Batch 2 with DIFFERENT pooled embeddings per sample: the time-embedding row vector is per batch sample."""
    cfgs, fam = tiny_xl
    e = _unet_cn_case(cfgs, fam, dev, dtype, 2, 8, 8)
    assert max(e) < _limits(dtype), e


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unet_controlnet_tiny_xl_nonsquare(dev, tiny_xl, dtype):
    cfgs, fam = tiny_xl
    e = _unet_cn_case(cfgs, fam, dev, dtype, 1, 8, 24)
    assert max(e) < _limits(dtype), e


def _pipeline_case(cfgs, fam, dev, dtype, hh, ww, steps, nimg, upcast=True, guidance=0.0, negative=False, sampler="ddim"):
    from oracle.canny import generate_canny_array
    ids1, ids2 = _ids(cfgs, nimg)
    n1 = n2 = None
    if negative:
        n1, n2 = _ids(cfgs, 1, seed=9)
    ctrls = np.stack([generate_canny_array(synthetic_image(hh, ww, 10 + i), 120, 200) for i in range(nimg)])
    g = torch.manual_seed(1)
    lat = torch.cat([torch.randn((1, 4, hh // 8, ww // 8), generator=g, dtype=torch.float32) for _ in range(nimg)])
    tn = (lambda a: None if a is None else torch.from_numpy(a))
    refs = [OP.sdxl_controlnet_pipeline(fam, cfgs, torch.from_numpy(ids1[i:i + 1]), torch.from_numpy(ids2[i:i + 1]), ctrls[i],
                                        lat[i:i + 1], steps, return_latents=True, guidance_scale=guidance, neg_ids1=tn(n1),
                                        neg_ids2=tn(n2), sampler=sampler) for i in range(nimg)]
    pipe = StableDiffusionXLControlNetPipeline(fam, cfgs)
    if sampler == "unipc":
        from saspa_aug_amd.scheduler import UniPCMultistepScheduler
        pipe.scheduler = UniPCMultistepScheduler.from_config(pipe.scheduler.config)
    if upcast:
        pipe.upcast_vae()
    pipe = pipe.to(dev, dtype)
    out, x, img = pipe.generate_batch(ids1, n1, ctrls, lat, steps, guidance, 0.75, return_latents=True, negative_ids_2=n2)
    ref_u8 = np.concatenate([r[0] for r in refs])
    ref_img = torch.cat([r[2] for r in refs])
    ref_x = torch.cat([r[1] for r in refs])
    got_img = from_nhwc(img, 3)
    d01 = ((got_img / 2 + 0.5).clamp(0, 1) - (ref_img / 2 + 0.5).clamp(0, 1)).abs().max().item()
    du8 = np.abs(out.cpu().numpy().astype(int) - ref_u8.astype(int)).max()
    ex = _relerr(from_nhwc(x, 4), ref_x)
    return d01, int(du8), ex, pipe


def test_sdxl_pipeline_fp32_parity_tiny(dev, tiny_xl):
    """North-star bar on the reference's SDXL-Turbo operating point: 2 DDIM steps, no CFG, conditioning scale 0.75."""
    cfgs, fam = tiny_xl
    d01, du8, ex, _ = _pipeline_case(cfgs, fam, dev, torch.float32, 64, 64, 2, nimg=2)
    assert d01 < 1e-3 and du8 <= 1 and ex < 3e-4, (d01, du8, ex)
    d01, du8, ex, _ = _pipeline_case(cfgs, fam, dev, torch.float32, 64, 128, 4, nimg=1)
    assert d01 < 1e-3 and du8 <= 1 and ex < 3e-4, (d01, du8, ex)


@pytest.mark.parametrize("graph", ["1", "0"])
def test_sdxl_pipeline_unipc_fp32_parity_tiny(dev, tiny_xl, graph, monkeypatch):
    """sampler = "unipcmultistep" on sd_xl-turbo (run_aug/run_aug.py:223-226): UniPC built from the pipeline's scheduler
    config ("trailing" spacing: 2 steps -> t = 999, 499), no CFG (the CFG-free update kernel saspa_unipc_step), and with
    CFG through the same path; fp32 vs the oracle, graph replay and the Python launch loop."""
    monkeypatch.setenv("SASPA_GRAPH", graph)
    from saspa_aug_amd.scheduler import UniPCMultistepScheduler
    from saspa_aug_amd.scheduler import SDXL_TURBO_SCHEDULER_CONFIG
    sch = UniPCMultistepScheduler.from_config(SDXL_TURBO_SCHEDULER_CONFIG)
    assert list(sch.set_timesteps(2)) == [999, 499] and list(sch.set_timesteps(4)) == [999, 749, 499, 249]
    cfgs, fam = tiny_xl
    d01, du8, ex, pipe = _pipeline_case(cfgs, fam, dev, torch.float32, 64, 64, 2, nimg=2, sampler="unipc")
    assert isinstance(pipe.scheduler, UniPCMultistepScheduler)
    assert d01 < 1e-3 and du8 <= 1 and ex < 3e-4, (d01, du8, ex)
    d01, du8, ex, _ = _pipeline_case(cfgs, fam, dev, torch.float32, 64, 128, 4, nimg=1, sampler="unipc")
    assert d01 < 1e-3 and du8 <= 1 and ex < 3e-4, (d01, du8, ex)
    d01, du8, ex, _ = _pipeline_case(cfgs, fam, dev, torch.float32, 64, 64, 4, nimg=2, guidance=5.0, negative=True, sampler="unipc")
    assert d01 < 1e-3 and du8 <= 1 and ex < 5e-4, (d01, du8, ex)


@pytest.mark.parametrize("negative", [False, True])
def test_sdxl_pipeline_cfg_fp32_parity_tiny(dev, tiny_xl, negative, monkeypatch):
    """Classifier-free guidance on the SDXL pipeline (BASELINE configs[4] family: guidance on, 4 steps): uncond half
    first; no negative prompt -> zero embeddings (force_zeros_for_empty_prompt), else both towers encode it.  fp32 path vs
    the oracle, graph replay and the Python launch loop."""
    cfgs, fam = tiny_xl
    for graph in ("1", "0"):
        monkeypatch.setenv("SASPA_GRAPH", graph)
        d01, du8, ex, _ = _pipeline_case(cfgs, fam, dev, torch.float32, 64, 64, 4, nimg=2, guidance=5.0, negative=negative)
        assert d01 < 1e-3 and du8 <= 1 and ex < 5e-4, (graph, d01, du8, ex)


def test_sdxl_pipeline_bf16_with_fp32_vae(dev, tiny_xl):
    """Production path: bf16 denoiser, VAE upcast to fp32 (run_aug/run_aug.py:224); latents within bf16 noise of the
    oracle's, run deterministic."""
    cfgs, fam = tiny_xl
    d01, du8, ex, pipe = _pipeline_case(cfgs, fam, dev, torch.bfloat16, 64, 64, 2, nimg=2)
    assert pipe.vae.dtype == torch.float32 and pipe.unet.dtype == torch.bfloat16
    assert ex < 8e-2, (d01, du8, ex)
    ids1, _ = _ids(cfgs, 2)
    ctrl = np.zeros((2, 64, 64, 3), np.uint8)
    lat = torch.randn((2, 4, 8, 8), generator=torch.manual_seed(5))
    a = pipe.generate_batch(ids1, None, ctrl, lat, 2)
    b = pipe.generate_batch(ids1, None, ctrl, lat, 2)
    assert torch.equal(a, b)
    # upcast_vae() after .to() re-packs the decoder from the retained state dict
    p2 = StableDiffusionXLControlNetPipeline(fam, cfgs).to(dev, torch.float16)
    assert p2.vae.dtype == torch.bfloat16
    p2.upcast_vae()
    assert p2.vae.dtype == torch.float32
    assert torch.equal(p2.generate_batch(ids1, None, ctrl, lat, 2), a)


def test_sdxl_call_form_and_init_pipeline(dev, tiny_xl):
    """The reference's construction + call: init_pipeline("sd_xl-turbo", "canny", 0) -> DDIM trailing + upcast VAE;
    pipe(prompt, image=PIL, num_inference_steps=2, generator, guidance_scale=0, negative_prompt=None, ...)."""
    from PIL import Image
    from saspa_aug_amd import run_aug as R
    from saspa_aug_amd.scheduler import DDIMScheduler
    cfgs, fam = tiny_xl
    pipe = R.init_pipeline("sd_xl-turbo", "canny", 0, cfgs=cfgs, state_dicts=fam)
    assert isinstance(pipe, StableDiffusionXLControlNetPipeline) and isinstance(pipe.scheduler, DDIMScheduler)
    assert pipe.scheduler.config["timestep_spacing"] == "trailing" and pipe._vae_fp32
    pipe = pipe.to("cuda:0", torch.float16)
    ctrl = Image.fromarray(np.zeros((64, 96, 3), np.uint8))
    kw = dict(prompt="a bird on a branch", image=ctrl, num_inference_steps=2, guidance_scale=0, negative_prompt=None,
              controlnet_conditioning_scale=0.75)
    a = R.pass_thorugh_pipe("sd_xl-turbo", pipe, "a bird on a branch", None, 0, 0.85, 2, torch.manual_seed(1), 0, 0.75,
                            negative_prompt=None, control_image=ctrl)
    b = pipe(generator=torch.manual_seed(1), **kw).images[0]
    assert a.size == (96, 64) and np.array_equal(np.asarray(a), np.asarray(b))
    c = pipe(generator=torch.manual_seed(1), **dict(kw, guidance_scale=5.0, negative_prompt="blurry")).images[0]
    assert c.size == (96, 64) and not np.array_equal(np.asarray(c), np.asarray(b))      # guidance changes the image
    # sampler = "unipcmultistep" (run_aug/run_aug.py:223-226): same construction, UniPC from the pipeline's config
    from saspa_aug_amd.scheduler import UniPCMultistepScheduler
    pu = R.init_pipeline("sd_xl-turbo", "canny", 0, sampler="unipcmultistep", cfgs=cfgs, state_dicts=fam)
    assert isinstance(pu.scheduler, UniPCMultistepScheduler) and pu.scheduler.config["timestep_spacing"] == "trailing"
    d = pu.to("cuda:0", torch.float16)(generator=torch.manual_seed(1), **kw).images[0]
    assert d.size == (96, 64) and not np.array_equal(np.asarray(d), np.asarray(b))


@pytest.fixture(scope="module")
def full_xl():
    """Full-width SDXL UNet + ControlNet state dicts (3.8 B parameters, ~40 s to synthesise) and the oracle's evaluation,
    shared by the dtype variants; released at the end of the module."""
    cfgs = CFG.SDXL_TURBO
    fam = dict(unet=W.synth_state_dict("unet", cfgs["unet"], 0), controlnet=W.synth_state_dict("controlnet", cfgs["controlnet"], 1))
    cache = {}
    yield cfgs, fam, cache
    fam.clear()
    cache.clear()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sdxl_full_width_step(dev, full_xl, dtype):
    """One SDXL UNet (2.57 B parameters) + ControlNet (1.25 B) evaluation at full width on a 128x128 image (16x16
    latents): every real channel count, head dim 64 with 5/10/20 heads, transformer depths 2 and 10, 2048-wide context."""
    cfgs, fam, cache = full_xl
    e = _unet_cn_case(cfgs, fam, dev, dtype, 1, 16, 16, cache=cache)
    assert max(e) < (1e-3 if dtype == torch.float32 else 0.12), e

"""GPU parity of the safety-checker row: Pillow-exact bicubic resize + crop + normalise on the device,
CLIP ViT-L/14 vision tower, concept similarities and the black-out, against the oracle (oracle/image_ops.py, whose resize
is itself pinned bit for bit against Pillow in tests/test_oracle.py) and against Pillow directly."""
import numpy as np
import pytest
import torch
from PIL import Image

import saspa_aug_amd  # noqa: F401
from oracle import image_ops as IO
from oracle import pipeline as OP
from saspa_aug_amd import config as CFG
from saspa_aug_amd import imageproc as IP
from saspa_aug_amd import models
from saspa_aug_amd import weights as W
from saspa_aug_amd.synthetic import synthetic_image

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w,oh,ow", [(512, 512, 224, 224), (512, 704, 224, 308), (300, 200, 336, 224), (64, 96, 224, 336),
                                       (37, 53, 20, 91), (224, 224, 224, 224), (512, 512, 224, 512)])
def test_resize_bicubic_is_pillow_exact(dev, h, w, oh, ow):
    rs = np.random.RandomState(h * 7 + w)
    imgs = np.stack([rs.randint(0, 256, (h, w, 3)).astype(np.uint8), synthetic_image(h, w, 3)])
    ref = np.stack([np.asarray(Image.fromarray(im).resize((ow, oh), Image.BICUBIC)) for im in imgs])
    got = IP.resize_bicubic_u8(torch.from_numpy(imgs).to(dev), oh, ow).cpu().numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)
    # a crop folded into the tables == cropping Pillow's result
    top, left, ch, cw = oh // 5, ow // 7, oh // 2, ow // 2
    got_c = IP.resize_bicubic_u8(torch.from_numpy(imgs).to(dev), oh, ow, crop=(top, left, ch, cw)).cpu().numpy()
    assert np.array_equal(got_c, ref[:, top:top + ch, left:left + cw])


@pytest.mark.parametrize("h,w", [(512, 512), (512, 704), (640, 512), (100, 260)])
def test_clip_image_preprocess_matches_oracle(dev, h, w):
    img = synthetic_image(h, w, 5)
    ref = IO.clip_image_preprocess(img)                                       # [3,224,224]
    got = IP.clip_image_preprocess(torch.from_numpy(img[None]).to(dev), torch.float32)[0].cpu()
    assert got.shape == (224, 224, 8) and (got[..., 3:] == 0).all()
    assert torch.equal(got[..., :3].permute(2, 0, 1), ref)                    # integer resize exact, same fp32 arithmetic


def _case(cfg, sd, dev, dtype, n=3, size=None):
    size = size or 4 * cfg["image_size"]
    imgs = np.stack([synthetic_image(size, size, 20 + i) for i in range(n)])
    px = torch.stack([IO.clip_image_preprocess(im, cfg["image_size"]) for im in imgs])
    flags, cs, ss = IO.safety_checker_forward(sd, cfg, px)
    net = models.SafetyChecker(sd, cfg, dev, dtype)
    pxd = IP.clip_image_preprocess(torch.from_numpy(imgs).to(dev), dtype, cfg["image_size"])
    gf, gcs, gss = net.decide(*net.similarity(pxd))
    return imgs, net, (flags, cs, ss), (gf, np.asarray(gcs), np.asarray(gss))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_safety_checker_tiny(dev, dtype):
    cfg = CFG.tiny()["safety"]
    sd = W.synth_state_dict("safety", cfg, seed=9)
    imgs, net, (flags, cs, ss), (gf, gcs, gss) = _case(cfg, sd, dev, dtype)
    lim = 2e-3 if dtype == torch.float32 else 6e-2           # scores are rounded to 3 decimals upstream
    assert np.abs(gcs - cs).max() <= lim and np.abs(gss - ss).max() <= lim, (np.abs(gcs - cs).max(), np.abs(gss - ss).max())
    assert not any(flags) and gf == flags                    # synthetic thresholds sit 6 sigma out
    out, f2 = net.forward(torch.from_numpy(imgs).to(dev))
    assert [bool(v) for v in f2.cpu().tolist()] == flags and np.array_equal(out.cpu().numpy(), imgs)


def test_safety_checker_blacks_out_flagged_images(dev):
    """Thresholds moved into the bulk of the score distribution: some images are flagged, the special-care adjustment
    fires, and the flagged images come back black -- all equal to the oracle's decisions (fp32 path)."""
    cfg = CFG.tiny()["safety"]
    sd = W.synth_state_dict("safety", cfg, seed=9)
    imgs, net, (_, cs, ss), _ = _case(cfg, sd, dev, torch.float32, n=6)
    sd = dict(sd)
    # thresholds midway between sorted per-image maxima so decisions are not borderline
    raw_c = cs + sd["concept_embeds_weights"].double().numpy()[None]
    mx = np.sort(raw_c.max(1))
    thr = float((mx[2] + mx[3]) / 2)
    sd["concept_embeds_weights"] = torch.full_like(sd["concept_embeds_weights"], thr)
    sd["special_care_embeds_weights"] = torch.full_like(sd["special_care_embeds_weights"], 10.0)
    ref_out, ref_flags = OP.run_safety_checker(sd, cfg, imgs)
    assert 0 < sum(ref_flags) < len(ref_flags)
    net = models.SafetyChecker(sd, cfg, dev, torch.float32)
    out, flags = net.forward(torch.from_numpy(imgs).to(dev))          # decision + black-out on the device
    flags = [bool(v) for v in flags.cpu().tolist()]
    host_flags, _, _ = net.decide(*net.similarity(IP.clip_image_preprocess(torch.from_numpy(imgs).to(dev), torch.float32,
                                                                           cfg["image_size"])))
    assert flags == ref_flags == host_flags and np.array_equal(out.cpu().numpy(), ref_out)
    assert all((out[i] == 0).all() for i, f in enumerate(flags) if f)
    # special-care concept firing adds 0.01 to every concept score of that image
    sd["special_care_embeds_weights"] = torch.full_like(sd["special_care_embeds_weights"], -10.0)
    px = torch.stack([IO.clip_image_preprocess(im, cfg["image_size"]) for im in imgs])
    _, cs2, _ = IO.safety_checker_forward(sd, cfg, px)
    net2 = models.SafetyChecker(sd, cfg, dev, torch.float32)
    pxd = IP.clip_image_preprocess(torch.from_numpy(imgs).to(dev), torch.float32, cfg["image_size"])
    hf2, gcs2, _ = net2.decide(*net2.similarity(pxd))
    assert np.abs(np.asarray(gcs2) - cs2).max() <= 2e-3
    # device decision with the adjustment on == the host replay of upstream's loop
    _, df2 = net2.forward(torch.from_numpy(imgs).to(dev))
    assert [bool(v) for v in df2.cpu().tolist()] == hf2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_safety_checker_full_width(dev, dtype):
    """The real shape: CLIP ViT-L/14 vision tower (24 layers, 1024 wide, 257 tokens), 303 981 588 parameters."""
    cfg = CFG.SAFETY_CHECKER
    sd = W.synth_state_dict("safety", cfg, seed=2)
    assert sum(v.numel() for v in sd.values()) == 303981588
    imgs, net, (flags, cs, ss), (gf, gcs, gss) = _case(cfg, sd, dev, dtype, n=2, size=512)
    lim = 2e-3 if dtype == torch.float32 else 3e-2
    assert np.abs(gcs - cs).max() <= lim and np.abs(gss - ss).max() <= lim, (np.abs(gcs - cs).max(), np.abs(gss - ss).max())
    assert gf == flags == [False, False]


def test_pipeline_runs_the_checker_and_can_disable_it(dev):
    from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline
    cfgs = CFG.tiny()
    fam = W.synth_family(cfgs, seed=3)
    pipe = StableDiffusionControlNetPipeline(fam, cfgs).to(dev, torch.float16)
    assert pipe.safety_checker is not None
    ctrl = Image.fromarray(np.zeros((64, 64, 3), np.uint8))
    r = pipe(prompt="an airplane", image=ctrl, num_inference_steps=2, generator=torch.manual_seed(1), negative_prompt="x",
             controlnet_conditioning_scale=0.75)
    assert r.nsfw_content_detected == [False]
    pipe.safety_checker = None
    r2 = pipe(prompt="an airplane", image=ctrl, num_inference_steps=2, generator=torch.manual_seed(1), negative_prompt="x",
              controlnet_conditioning_scale=0.75)
    assert r2.nsfw_content_detected is None and np.array_equal(np.asarray(r.images[0]), np.asarray(r2.images[0]))

"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the image pre-processing and of the safety checker
(StableDiffusionSafetyChecker) the reference's SD pipeline runs after decoding
([upstream] diffusers 0.32.2 pipelines/stable_diffusion/safety_checker.py + transformers CLIPImageProcessor /
CLIPVisionModel, recalled; the reference never disables the checker, run_aug/run_aug.py:185-207).

The resize restates Pillow's ImagingResample for 8-bit channels (src/libImaging/Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc) and IS PINNED: tests compare it bit for bit
with the Pillow installed here (tests/test_oracle.py).  The network part is parity unpinned like the rest of oracle/."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import sd_models as M

PRECISION_BITS = 32 - 8 - 2
CLIP_IMAGE_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_IMAGE_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_filter(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size, support=2.0, filt=bicubic_filter):
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.float64)
    bounds = np.zeros((out_size, 2), np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size)
        xmax -= xmin
        for x in range(xmax):
            w = filt((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    ik = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << PRECISION_BITS)), np.trunc(0.5 + kk * (1 << PRECISION_BITS))).astype(np.int64)
    return bounds, ik


def _pass(img, out_size, axis):
    """img u8 [H,W,C]; resample along axis (0 vertical, 1 horizontal)."""
    src = np.moveaxis(img.astype(np.int64), axis, 0)              # [L, ...]
    bounds, ik = precompute_coeffs(src.shape[0], out_size)
    out = np.empty((out_size,) + src.shape[1:], np.int64)
    for t in range(out_size):
        lo, cnt = bounds[t]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for k in range(cnt):
            acc += src[lo + k] * ik[t, k]
        out[t] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resize_bicubic_u8(img, out_h, out_w):
    """PIL.Image.fromarray(img).resize((out_w, out_h), PIL.Image.BICUBIC): horizontal pass, then vertical."""
    x = img
    if out_w != img.shape[1]:
        x = _pass(x, out_w, 1)
    if out_h != img.shape[0]:
        x = _pass(x, out_h, 0)
    return x


def clip_image_preprocess(img, size=224):
    """CLIPImageProcessor(size=224 shortest edge, bicubic, center crop 224, rescale 1/255, CLIP mean/std):
    u8 [H,W,3] -> fp32 [3,size,size]."""
    h, w = img.shape[:2]
    oh, ow = (size, int(size * w / h)) if h <= w else (int(size * h / w), size)
    x = resize_bicubic_u8(img, oh, ow)
    top, left = (oh - size) // 2, (ow - size) // 2
    x = x[top:top + size, left:left + size]
    x = (x.astype(np.float64) * (1 / 255.0)).astype(np.float32)
    x = (x - np.asarray(CLIP_IMAGE_MEAN, np.float32)) / np.asarray(CLIP_IMAGE_STD, np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))


def clip_vision_forward(sd, cfg, pixel_values, pfx="vision_model.vision_model"):
    """CLIPVisionModel (ViT-L/14 for the safety checker): returns pooled_output = post_layernorm(last_hidden[:, 0]).
    Key names as in the checkpoint ("pre_layrnorm" sic)."""
    b = pixel_values.shape[0]
    x = F.conv2d(pixel_values, sd[pfx + ".embeddings.patch_embedding.weight"], None, stride=cfg["patch"])
    x = x.flatten(2).transpose(1, 2)
    cls = sd[pfx + ".embeddings.class_embedding"].expand(b, 1, -1)
    x = torch.cat([cls, x], 1) + sd[pfx + ".embeddings.position_embedding.weight"][None]
    x = M.layer_norm(sd, pfx + ".pre_layrnorm", x)
    for i in range(cfg["layers"]):
        lp = f"{pfx}.encoder.layers.{i}"
        h = M.layer_norm(sd, lp + ".layer_norm1", x)
        q = M.linear(sd, lp + ".self_attn.q_proj", h)
        k = M.linear(sd, lp + ".self_attn.k_proj", h)
        v = M.linear(sd, lp + ".self_attn.v_proj", h)
        x = x + M.linear(sd, lp + ".self_attn.out_proj", M.mha(q, k, v, cfg["heads"]))
        h = M.layer_norm(sd, lp + ".layer_norm2", x)
        h = M.linear(sd, lp + ".mlp.fc1", h)
        h = h * torch.sigmoid(1.702 * h)
        x = x + M.linear(sd, lp + ".mlp.fc2", h)
    return M.layer_norm(sd, pfx + ".post_layernorm", x[:, 0])


def safety_checker_forward(sd, cfg, clip_input):
    """StableDiffusionSafetyChecker.forward on pre-processed pixels [B,3,224,224] -> (has_nsfw [B] bool list,
    concept scores [B,17], special scores [B,3]) with the per-image adjustment logic (0.01 once a special-care
    concept fires) and round(..., 3)."""
    pooled = clip_vision_forward(sd, cfg, clip_input)
    emb = F.linear(pooled, sd["visual_projection.weight"])

    def cos(a, b):
        return F.normalize(a) @ F.normalize(b).t()
    special = cos(emb, sd["special_care_embeds"]).double().numpy()
    concept = cos(emb, sd["concept_embeds"]).double().numpy()
    sw = sd["special_care_embeds_weights"].double().numpy()
    cw = sd["concept_embeds_weights"].double().numpy()
    flags, cs, ss = [], [], []
    for i in range(emb.shape[0]):
        adj = 0.0
        s_scores = []
        for j in range(len(sw)):                      # upstream: the adjustment switches on INSIDE this loop
            s_scores.append(round(float(special[i, j] - sw[j] + adj), 3))
            if s_scores[-1] > 0:
                adj = 0.01
        c_scores = [round(float(concept[i, j] - cw[j] + adj), 3) for j in range(len(cw))]
        flags.append(any(v > 0 for v in c_scores))
        cs.append(c_scores)
        ss.append(s_scores)
    return flags, np.asarray(cs), np.asarray(ss)

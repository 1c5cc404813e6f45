"""The filter stage that follows generation: the two models the reference runs inside
`create_json_of_image_name_to_augmented_images_paths` (all_utils/utils.py:252-255, :306-323, :357-375), as launch
sequences of the gfx950 kernels.

  * `SemanticFilter` -- OpenAI CLIP RN50 (`clip.load('RN50')`, all_utils/utils.py:253): image tower = ModifiedResNet +
    attention pool, text tower = 12-layer transformer; an augmented image passes when the dataset's positive prompt
    (`ds_utils.get_basic_prompt()`) beats the six negative prompts (:306-312, `get_semantic_filtering` :169-177).
  * `ConfidenceFilter` -- the baseline classifier WSDAN_CAL (fgvc/models/cal.py:131-228; loader
    all_utils/dataset_utils.py:87-115): passes when the SOURCE image's label is among the top-k (10) logits (:357-366).

MI355X-first choices: BatchNorm (inference) is folded into the conv weights + a bias at pack time, ReLU rides in the GEMM
epilogue (`SASPA_ACT_RELU`, `SASPA_ACT_ADD_RELU` for the bottleneck's add-then-ReLU), pooling is one streaming kernel,
pre-processing (PIL-exact bicubic / bilinear resize, crop, normalise) runs on the device from the decoded u8 image, images
are processed in batches.  Both models run on the exact-fp32 MFMA path by default: the whole stage is ~12 GFLOP per image
(0.01 % of generating it), and an argmax / top-k decision should not move with bf16 rounding.

No CPU or eager-PyTorch arithmetic: torch is used for device memory and layout copies (cat / transpose) only."""
import logging
import os
from pathlib import Path

import numpy as np
import torch

from . import imageproc, models, ops
from . import weights as W
from .config import CLIP_RN50, WSDAN_CAL_R50, WSDAN_CAL_R101

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
NEGATIVE_PROMPTS = ["a photo of an object", "a photo of a scene", "a photo of geometric shapes", "a photo", "an image",
                    "a black photo"]                                     # all_utils/utils.py:307
RELU, ADD_RELU = ops.ACT_RELU, ops.ACT_ADD_RELU


def _center_crop_origin(length, size):
    return int(round((length - size) / 2.0))        # torchvision.transforms.functional.center_crop (Python round)


def rn50_preprocess(src_u8, dtype, size=224):
    """clip._transform(n_px): Resize(n_px, BICUBIC) of the shorter side, CenterCrop(n_px), /255, CLIP mean / std, on a
    device u8 [n,H,W,3] batch -> [n,size,size,8]."""
    n, h, w, _ = src_u8.shape
    if h <= w:
        oh, ow = size, int(size * w / h)
    else:
        oh, ow = int(size * h / w), size
    top, left = _center_crop_origin(oh, size), _center_crop_origin(ow, size)
    x = imageproc.resize_u8(src_u8.contiguous(), oh, ow, crop=(top, left, size, size), filt="bicubic")
    return imageproc.normalize_u8(x, dtype)


def cal_preprocess(src_u8, dtype, size=224):
    """BaseUtils.get_transform (all_utils/dataset_utils.py:77-85): Resize((size/0.875,)*2) bilinear, CenterCrop(size),
    ToTensor, Normalize(ImageNet)."""
    big = int(size / 0.875)
    o = _center_crop_origin(big, size)
    x = imageproc.resize_u8(src_u8.contiguous(), big, big, crop=(o, o, size, size), filt="bilinear")
    return imageproc.normalize_u8(x, dtype, IMAGENET_MEAN, IMAGENET_STD)


class _Convs:
    """Bias-free convs with their BatchNorm folded in, packed for the implicit-GEMM kernel."""

    def __init__(self, sd, dev, dtype):
        self.sd, self.dev, self.dtype, self.p = sd, dev, dtype, {}

    def conv_bn(self, name, conv, bn, eps=1e-5):
        w, b = W.fold_bn(self.sd[conv + ".weight"], self.sd, bn, eps)
        kh, kw = w.shape[2], w.shape[3]
        pk = W.pack_conv(w)
        chunk = W.chunk_major_ok(kh, kw, W.round8(w.shape[1]), 0, self.dtype)
        if chunk:
            pk = W.to_chunk_major(pk, kh * kw, self.dtype)
        t = pk.to(self.dev, self.dtype)
        t.saspa_korder = 1 if chunk else 0
        self.p[name + ".w"], self.p[name + ".b"], self.p[name + ".k"] = t, b.to(self.dev, torch.float32).contiguous(), (kh, kw)

    def run(self, name, x, stride=1, pad=0, act=RELU, residual=None):
        kh, kw = self.p[name + ".k"]
        return ops.conv(x, self.p[name + ".w"], self.p[name + ".b"], kh=kh, kw=kw, stride=stride, pad=pad, act=act,
                        residual=residual)


class ClipRN50Visual:
    """clip.model.ModifiedResNet + AttentionPool2d (keys `visual.*` of the OpenAI checkpoint)."""

    def __init__(self, sd, cfg, dev, dtype=torch.float32):
        self.cfg, self.dev, self.dtype = cfg, dev, dtype
        c = self.c = _Convs(sd, dev, dtype)
        v = "visual"
        for i in (1, 2, 3):
            c.conv_bn(f"stem{i}", f"{v}.conv{i}", f"{v}.bn{i}")
        self.blocks = []
        for li, nb in enumerate(cfg["layers"]):
            for bi in range(nb):
                pf = f"{v}.layer{li + 1}.{bi}"
                for j in (1, 2, 3):
                    c.conv_bn(f"{pf}.c{j}", f"{pf}.conv{j}", f"{pf}.bn{j}")
                ds = pf + ".downsample.0.weight" in sd
                if ds:
                    c.conv_bn(f"{pf}.ds", f"{pf}.downsample.0", f"{pf}.downsample.1")
                self.blocks.append((pf, 2 if (bi == 0 and li > 0) else 1, ds))
        a = v + ".attnpool"
        pos = sd[a + ".positional_embedding"].double()
        p = self.p = {}
        for n in ("q", "k", "v"):
            wt, bs = sd[f"{a}.{n}_proj.weight"], sd[f"{a}.{n}_proj.bias"]
            p[n + ".w"] = wt.contiguous().to(dev, dtype)
            # (t + pos) @ W^T + b = t @ W^T + (pos @ W^T + b): the position table becomes a per-token additive term
            p[n + ".r"] = (pos @ wt.double().t() + bs.double()).float().to(dev, dtype).contiguous()
        p["c.w"] = sd[a + ".c_proj.weight"].contiguous().to(dev, dtype)
        p["c.b"] = sd[a + ".c_proj.bias"].float().to(dev).contiguous()
        self.heads = cfg["heads"]
        c.sd = None

    def forward(self, pixels):
        """[B,S,S,8] normalised channels-last pixels -> [B, embed_dim] (dtype of the tower)."""
        c = self.c
        x = c.run("stem1", pixels, stride=2, pad=1)
        x = c.run("stem2", x, pad=1)
        x = c.run("stem3", x, pad=1)
        x = ops.pool2d(x, 2)
        for pf, stride, ds in self.blocks:
            out = c.run(pf + ".c1", x)
            out = c.run(pf + ".c2", out, pad=1)
            if stride > 1:
                out = ops.pool2d(out, stride)
            identity = x
            if ds:
                identity = c.run(pf + ".ds", ops.pool2d(x, stride) if stride > 1 else x, act=ops.ACT_NONE)
            x = c.run(pf + ".c3", out, act=ADD_RELU, residual=identity)
        b, hh, ww, ch = x.shape
        n = hh * ww
        mean = ops.pool2d(x, hh)                                             # the mean token (hh == ww)
        t = torch.cat([mean.view(b, 1, ch), x.view(b, n, ch)], 1).contiguous()   # [B, n+1, C]  (layout copy)
        p = self.p
        nt = n + 1

        def proj(name, rows):
            r = p[name + ".r"][:rows]
            return ops.linear(t[:, :rows].contiguous() if rows != nt else t, p[name + ".w"],
                              residual=r[None].expand(b, -1, -1).contiguous())
        q = proj("q", 1)                                                      # query = the mean token only
        k = proj("k", nt)
        vmat = proj("v", nt)
        ld = ops.round8(nt)
        vt = torch.zeros((b, ch, ld), device=x.device, dtype=x.dtype)
        vt[:, :, :nt] = vmat[:, :, :ch].transpose(1, 2)                       # keys contiguous (layout copy)
        o = models.attention_core(q, k, vt, self.heads, 1, nt)
        return ops.linear(o.view(b, ch), p["c.w"], p["c.b"])


class SemanticFilter:
    """CLIP_selector (all_utils/utils.py:137-166) with the prompts [positive] + NEGATIVE_PROMPTS; `passes(images)` is
    `get_semantic_filtering` for a batch: argmax over the prompts == 0."""

    def __init__(self, sd, cfg, dev, positive_prompt, tokenizer, dtype=torch.float32):
        self.cfg, self.dev, self.dtype = cfg, dev, dtype
        self.visual = ClipRN50Visual(sd, cfg, dev, dtype)
        tcfg = dict(vocab=cfg["vocab"], width=cfg["text_width"], layers=cfg["text_layers"], heads=cfg["text_heads"],
                    mlp=4 * cfg["text_width"], max_pos=cfg["context"])
        text = models.CLIPText(W.openai_clip_text_to_hf(sd, cfg["text_layers"]), tcfg, dev, dtype)
        self.prompts = [positive_prompt] + NEGATIVE_PROMPTS
        ids = np.concatenate([tokenizer(pr) for pr in self.prompts])          # [7, 77]
        ids_t = ops.h2d(torch.from_numpy(ids), dev)
        hidden = text.forward(ids_t)                                          # final-LN states [7, 77, width]
        eot = ids_t.argmax(dim=-1)                                            # clip: the EOT token has the highest id
        rows = hidden[torch.arange(len(self.prompts), device=dev), eot].contiguous()
        feats = ops.linear(rows.float(), text.p["text_projection.w"].float())[:, :cfg["embed_dim"]]
        # constant for the run: unit text embeddings (the image norm and logit_scale are common factors of the argmax)
        self.text_unit = torch.nn.functional.normalize(feats.double(), dim=-1).float().contiguous()
        del text

    @torch.no_grad()
    def logits(self, images_u8):
        """device u8 [n,H,W,3] -> fp32 [n, 7] cosine-similarity logits up to the common positive factor."""
        px = rn50_preprocess(images_u8, self.dtype, self.cfg["image_size"])
        e = self.visual.forward(px)[:, :self.cfg["embed_dim"]].float().contiguous()
        return ops.linear(e, self.text_unit)[:, :len(self.prompts)]

    def passes(self, images_u8):
        return np.argmax(self.logits(images_u8).cpu().numpy(), axis=-1) == 0       # the decision is host control flow


class WSDANCAL:
    """WSDAN_CAL.forward in eval mode -> p (logits [B, num_classes]); resnet features, 1x1 attention conv + BN + ReLU,
    bilinear attention pooling (GAP form), sign-sqrt, L2 normalise, fc(feature_matrix * 100)."""

    def __init__(self, sd, cfg, dev, dtype=torch.float32):
        self.cfg, self.dev, self.dtype = cfg, dev, dtype
        c = self.c = _Convs(sd, dev, dtype)
        c.conv_bn("stem", "features.0", "features.1")
        self.blocks = []
        strides = (1, 2, 2, 1)      # fgvc ResNet(..., stride=1): layer4 is NOT strided (features at 1/16, 14 x 14 for 224 crops)
        for li, nb in enumerate(cfg["layers"]):
            for bi in range(nb):
                pf = f"features.{4 + li}.{bi}"
                for j in (1, 2, 3):
                    c.conv_bn(f"{pf}.c{j}", f"{pf}.conv{j}", f"{pf}.bn{j}")
                ds = pf + ".downsample.0.weight" in sd
                if ds:
                    c.conv_bn(f"{pf}.ds", f"{pf}.downsample.0", f"{pf}.downsample.1")
                self.blocks.append((pf, strides[li] if bi == 0 else 1, ds))
        wa, ba = W.fold_bn(sd["attentions.conv.weight"], sd, "attentions.bn", 1e-3)        # BasicConv2d: BN eps 0.001
        self.att_w = wa[:, :, 0, 0].contiguous().to(dev, dtype)                           # [M, C]
        self.att_b = ba.to(dev, dtype).contiguous()                                       # per attention map
        self.fc_w = sd["fc.weight"].contiguous().to(dev, torch.float32)
        self.num_classes = self.fc_w.shape[0]
        c.sd = None

    @torch.no_grad()
    def forward(self, pixels):
        c = self.c
        x = c.run("stem", pixels, stride=2, pad=3)
        x = ops.pool2d(x, 3, 2, 1, mode="max")
        for pf, stride, ds in self.blocks:
            out = c.run(pf + ".c1", x)
            out = c.run(pf + ".c2", out, stride=stride, pad=1)
            identity = c.run(pf + ".ds", x, stride=stride, act=ops.ACT_NONE) if ds else x
            x = c.run(pf + ".c3", out, act=ADD_RELU, residual=identity)
        b, hh, ww, ch = x.shape
        hw, m = hh * ww, self.att_w.shape[0]
        ld = ops.round8(hw)
        feat = x.view(b, hw, ch)
        # attention maps TRANSPOSED (swapped GEMM operands, like the V^T projection): att_t[b] = relu(Wa @ feat_b^T + ba)
        att_t = torch.zeros((b, m, ld), device=x.device, dtype=x.dtype)
        bias_t = torch.zeros((b, m, ld), device=x.device, dtype=x.dtype)          # per-row bias as the GEMM's residual
        bias_t[:, :, :hw] = self.att_b[None, :, None]
        ops.gemm_batched(self.att_w, self.att_w.stride(0), (0, 0), feat, feat.stride(1), (feat.stride(0), 0), att_t, ld, (m * ld, 0),
                         m, hw, ch, b, 1, residual=bias_t, ldr=ld, act=ADD_RELU)
        feat_t = torch.zeros((b, ch, ld), device=x.device, dtype=x.dtype)
        feat_t[:, :, :hw] = feat.transpose(1, 2)                                           # layout copy
        # feature_matrix[b] = att_t[b] @ feat_t[b]^T / HW  -> [M, C]
        fm = torch.empty((b, m, ch), device=x.device, dtype=x.dtype)
        ops.gemm_batched(att_t, ld, (m * ld, 0), feat_t, ld, (ch * ld, 0), fm, ch, (m * ch, 0), m, ch, ld, b, 1, alpha=1.0 / hw)
        fm = ops.signsqrt_l2norm(fm.view(b, m * ch).float(), 1e-6, 100.0)
        return ops.linear(fm, self.fc_w)[:, :self.num_classes]


class ConfidenceFilter:
    def __init__(self, sd, cfg, dev, top_k=10, dtype=torch.float32):
        self.cfg, self.dev, self.dtype = cfg, dev, dtype
        self.model = WSDANCAL(sd, cfg, dev, dtype)
        self.top_k = min(int(top_k), self.model.num_classes)                  # all_utils/utils.py:319

    @torch.no_grad()
    def logits(self, images_u8):
        return self.model.forward(cal_preprocess(images_u8, self.dtype, self.cfg["image_size"]))

    def passes(self, images_u8, labels):
        """`correct_label in logits.topk(k)[1]` per image (all_utils/utils.py:363-364); labels: ints, one per image."""
        lg = self.logits(images_u8).float().cpu().numpy()                      # the decision is host control flow
        return np.array([int((row > row[int(lb)]).sum()) < self.top_k for lb, row in zip(labels, lg)], dtype=bool)


# ---------------------------------------------------------------------------------------------------------------------
# checkpoints
# ---------------------------------------------------------------------------------------------------------------------
def load_cal_checkpoint(path):
    """A baseline-classifier checkpoint as fgvc/train.py saves it ({'state_dict': ...}; keys possibly prefixed with
    `_orig_mod.` by torch.compile, all_utils/dataset_utils.py:99-104) -> (state dict, config)."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    sd = ck["state_dict"] if "state_dict" in ck else ck
    sd = {k.replace("_orig_mod.", "").replace("module.", ""): v.float() for k, v in sd.items()
          if torch.is_tensor(v) and not k.endswith("num_batches_tracked")}
    n23 = any(k.startswith("features.6.22.") for k in sd)
    cfg = dict(WSDAN_CAL_R101 if n23 else WSDAN_CAL_R50, num_classes=sd["fc.weight"].shape[0])
    return sd, cfg


def load_clip_rn50(path):
    """OpenAI's RN50.pt (a TorchScript archive) or a plain state-dict file -> fp32 state dict."""
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except RuntimeError:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    return {k: v.float() for k, v in sd.items() if torch.is_tensor(v)}


def synthetic_filters_allowed():
    """Opt-in for architecture-exact SYNTHETIC filter weights (tests, benchmarks): SASPA_SYNTHETIC_FILTERS=1."""
    return os.environ.get("SASPA_SYNTHETIC_FILTERS", "0") not in ("", "0")


def filter_checkpoints(ds_utils, weights_dir, semantic=True, confidence=True):
    """Paths of the checkpoints the enabled filters need: (clip RN50 path | None, baseline checkpoint path | None).
    Raises FileNotFoundError for a missing / ambiguous checkpoint unless synthetic filter weights were asked for
    explicitly -- a filtered aug.json must never reflect the decisions of random models (the reference asserts exactly
    one baseline checkpoint, all_utils/dataset_utils.py:92, and loads the real CLIP, all_utils/utils.py:253)."""
    name = "compcars" if "compcars" in ds_utils.name else ds_utils.name
    rn = cp = None
    if semantic:
        cand = os.path.join(weights_dir, "clip", "RN50.pt") if weights_dir else None
        if cand and os.path.exists(cand):
            rn = cand
        elif not synthetic_filters_allowed():
            raise FileNotFoundError(
                f"semantic filter: {cand or '<WEIGHTS_DIR>/clip/RN50.pt'} not found.  Give WEIGHTS_DIR with the OpenAI CLIP RN50 "
                "checkpoint, set SEMANTIC_FILTERING = 0, or opt in to synthetic filter weights with SASPA_SYNTHETIC_FILTERS=1")
    if confidence:
        cdir = Path(weights_dir, "checkpoints", name) if weights_dir else None
        cps = sorted(cdir.glob("*.pth")) if cdir else []
        if len(cps) > 1:
            raise FileNotFoundError(f"Found {len(cps)} checkpoints in {cdir}. Expected 1")
        if cps:
            cp = str(cps[0])
        elif not synthetic_filters_allowed():
            raise FileNotFoundError(
                f"model-confidence filter: no baseline checkpoint (*.pth) in {cdir or '<WEIGHTS_DIR>/checkpoints/' + name}.  Train the "
                "baseline first (fgvc/train.py), set MODEL_CONFIDENCE_BASED_FILTERING = 0, or opt in to synthetic filter "
                "weights with SASPA_SYNTHETIC_FILTERS=1")
    return rn, cp


def build_filters(ds_utils, dev, semantic=True, confidence=True, weights_dir=None, top_k=10, tokenizer=None):
    """The filter models for a dataset.  `weights_dir` holds `clip/RN50.pt` and `checkpoints/<dataset>/*.pth` (the
    reference's `all_utils/checkpoints/<name>/`).  A missing checkpoint is an error (filter_checkpoints); only with
    SASPA_SYNTHETIC_FILTERS=1 does the stage run on architecture-exact SYNTHETIC weights, and says so loudly."""
    from .tokenizer import make_tokenizer
    sem = conf = None
    rn, cp = filter_checkpoints(ds_utils, weights_dir, semantic, confidence)
    if semantic:
        if rn:
            sd = load_clip_rn50(rn)
        else:
            logging.warning("semantic filter: SASPA_SYNTHETIC_FILTERS=1 -> SYNTHETIC CLIP-RN50 weights; its decisions are those of a random model")
            sd = W.synth_state_dict("clip_rn50", CLIP_RN50, 11)
        tok = tokenizer or make_tokenizer(os.path.join(weights_dir, "clip") if weights_dir else None, CLIP_RN50["vocab"], pad_id=0)
        sem = SemanticFilter(sd, CLIP_RN50, dev, ds_utils.get_basic_prompt(), tok)
    if confidence:
        if cp:
            sd, cfg = load_cal_checkpoint(cp)
        else:
            logging.warning("confidence filter: SASPA_SYNTHETIC_FILTERS=1 -> SYNTHETIC WSDAN_CAL (resnet101) weights; its decisions are those of a random model")
            cfg = dict(WSDAN_CAL_R101, num_classes=max(2, ds_utils.num_classes))
            sd = W.synth_state_dict("cal", cfg, 12)
        conf = ConfidenceFilter(sd, cfg, dev, top_k)
    return sem, conf


# ---------------------------------------------------------------------------------------------------------------------
# the stage itself (all_utils/utils.py:337-437, the filter part of the per-image loop)
# ---------------------------------------------------------------------------------------------------------------------
def _load_u8(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def _image_size(path):
    from PIL import Image
    with Image.open(path) as im:      # header only: nothing is decoded
        return im.size[1], im.size[0]


def apply_filters(mapping, original_images_paths, ds_utils, dev, semantic=None, confidence=None, batch_size=32):
    """mapping: {original file name: [augmented paths]} as `match_augmented_images` builds it (every original present).
    Returns (filtered mapping, counters).  Per original image the reference first drops the augmentations whose
    classifier top-k misses the source label, then those CLIP does not recognise as the meta class; both decisions are
    per augmented image and independent, so they are evaluated in batches (grouped by image size) and combined.
    Memory: a first pass reads only the PNG headers to group the paths by size; pixels are decoded per batch on a
    prefetch thread, so at most two batches are resident on the host (a real dataset has 13-16k augmentations of ~1 MB)."""
    from concurrent.futures import ThreadPoolExecutor
    if dev is None:
        for m in (semantic, confidence):
            if m is not None:
                dev = m.dev
                break
    labels = {}
    if confidence is not None:
        table = ds_utils.get_image_path_to_class_id_dict()
        for ip in original_images_paths:
            labels[Path(ip).name] = table[ip]
    work = [(name, ap) for name, aps in mapping.items() for ap in aps]
    keep = {}
    counters = dict(not_in_top_k=0, semantic=0)
    groups = {}
    for name, ap in work:
        groups.setdefault(_image_size(ap), []).append((name, ap))
    chunks = [items[i:i + batch_size] for _, items in sorted(groups.items()) for i in range(0, len(items), batch_size)]

    def decode(chunk):
        return np.stack([_load_u8(ap) for _, ap in chunk])

    with ThreadPoolExecutor(max_workers=1) as pool:
        nxt = pool.submit(decode, chunks[0]) if chunks else None
        for ci, chunk in enumerate(chunks):
            pixels = nxt.result()
            nxt = pool.submit(decode, chunks[ci + 1]) if ci + 1 < len(chunks) else None
            batch = ops.h2d(torch.from_numpy(pixels), dev)
            ok_c = confidence.passes(batch, [labels[name] for name, _ in chunk]) if confidence is not None else np.ones(len(chunk), bool)
            ok_s = semantic.passes(batch) if semantic is not None else np.ones(len(chunk), bool)
            for (name, ap), c_ok, s_ok in zip(chunk, ok_c, ok_s):
                if not c_ok:
                    counters["not_in_top_k"] += 1          # dropped first: never reaches the semantic filter (:357-366)
                elif not s_ok:
                    counters["semantic"] += 1
                keep[ap] = bool(c_ok and s_ok)
    out = {name: [ap for ap in aps if keep[ap]] for name, aps in mapping.items()}
    return out, counters

"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of the two filter models the reference runs inside
``create_json_of_image_name_to_augmented_images_paths`` (all_utils/utils.py:252-255, :306-323, :357-375, :401-409):

* semantic filter: OpenAI CLIP ``RN50`` (``clip.load('RN50')``, all_utils/utils.py:253; ``CLIP_selector`` :137-166): the
  image passes when the positive prompt (``get_basic_prompt()``) has the highest cosine similarity among
  [positive] + 6 negatives (``get_semantic_filtering`` :169-177).  Pre-processing = clip's ``_transform``:
  Resize(224, BICUBIC) on the PIL image (shorter side), CenterCrop(224) with torchvision's rounding, /255, CLIP mean / std.
* confidence filter: the baseline classifier ``WSDAN_CAL`` (fgvc/models/cal.py:131-228; ResNet-101 / -50 features
  fgvc/models/resnet.py:104-180, 1x1 attention conv + BN(eps 1e-3) + ReLU, bilinear attention pooling :43-83 in eval mode,
  ``fc(feature_matrix * 100)``): the image passes when the source image's label is among the top-k (10) logits
  (all_utils/utils.py:357-366).  Pre-processing = ``BaseUtils.get_transform`` (all_utils/dataset_utils.py:77-85):
  Resize((256, 256)) bilinear, CenterCrop(224), /255, ImageNet mean / std.

PARITY UNPINNED for CLIP RN50 (the ``clip`` package and its checkpoint are not available here: architecture and state-dict
key names follow the published OpenAI CLIP ``model.py``, recalled; the parameter count equals the public 102 007 137).
The CAL restatement IS PINNED: tests/golden/reference_filter_golden.json holds logits produced by the reference's own
``fgvc/models/cal.py`` (imported in the build container with torchvision stubbed, tests/golden/make_filter_golden.py) for
ResNet-50 and ResNet-101 on seeded weights / inputs; tests/test_filters_oracle.py checks this file against them.
"""
import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def bn(sd, pfx, x, eps=1e-5):
    return F.batch_norm(x, sd[pfx + ".running_mean"], sd[pfx + ".running_var"], sd[pfx + ".weight"], sd[pfx + ".bias"], False, 0.0, eps)


# ---------------------------------------------------------------------------------------------------------------------
# pre-processing (PIL, like torchvision's transforms on a PIL image)
# ---------------------------------------------------------------------------------------------------------------------
def _center_crop_box(h, w, size):
    top = int(round((h - size) / 2.0))           # torchvision.transforms.functional.center_crop
    left = int(round((w - size) / 2.0))
    return top, left


def rn50_preprocess(img_u8, size=224):
    """clip._transform(224): u8 [H,W,3] -> fp32 [3,224,224]."""
    im = Image.fromarray(img_u8)
    w, h = im.size
    if h <= w:
        oh, ow = size, int(size * w / h)
    else:
        oh, ow = int(size * h / w), size
    im = im.resize((ow, oh), Image.BICUBIC)
    top, left = _center_crop_box(oh, ow, size)
    a = np.asarray(im)[top:top + size, left:left + size]
    x = torch.from_numpy(a.copy()).float().div(255.0).permute(2, 0, 1)
    return (x - torch.tensor(CLIP_MEAN)[:, None, None]) / torch.tensor(CLIP_STD)[:, None, None]


def cal_preprocess(img_u8, resize=(224, 224)):
    """BaseUtils.get_transform: Resize((256,256)) bilinear -> CenterCrop(224) -> ToTensor -> Normalize(ImageNet)."""
    im = Image.fromarray(img_u8).resize((int(resize[1] / 0.875), int(resize[0] / 0.875)), Image.BILINEAR)
    top, left = _center_crop_box(im.size[1], im.size[0], resize[0])
    a = np.asarray(im)[top:top + resize[0], left:left + resize[1]]
    x = torch.from_numpy(a.copy()).float().div(255.0).permute(2, 0, 1)
    return (x - torch.tensor(IMAGENET_MEAN)[:, None, None]) / torch.tensor(IMAGENET_STD)[:, None, None]


# ---------------------------------------------------------------------------------------------------------------------
# OpenAI CLIP RN50
# ---------------------------------------------------------------------------------------------------------------------
def _clip_bottleneck(sd, pfx, x, stride):
    out = F.relu(bn(sd, pfx + ".bn1", F.conv2d(x, sd[pfx + ".conv1.weight"])))
    out = F.relu(bn(sd, pfx + ".bn2", F.conv2d(out, sd[pfx + ".conv2.weight"], padding=1)))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = bn(sd, pfx + ".bn3", F.conv2d(out, sd[pfx + ".conv3.weight"]))
    identity = x
    if pfx + ".downsample.0.weight" in sd:
        if stride > 1:
            identity = F.avg_pool2d(identity, stride)
        identity = bn(sd, pfx + ".downsample.1", F.conv2d(identity, sd[pfx + ".downsample.0.weight"]))
    return F.relu(out + identity)


def clip_rn50_visual(sd, cfg, x):
    """ModifiedResNet + AttentionPool2d: [B,3,224,224] -> [B, embed_dim]; keys under 'visual.'."""
    v = "visual"
    x = F.relu(bn(sd, v + ".bn1", F.conv2d(x, sd[v + ".conv1.weight"], stride=2, padding=1)))
    x = F.relu(bn(sd, v + ".bn2", F.conv2d(x, sd[v + ".conv2.weight"], padding=1)))
    x = F.relu(bn(sd, v + ".bn3", F.conv2d(x, sd[v + ".conv3.weight"], padding=1)))
    x = F.avg_pool2d(x, 2)
    for li, nblocks in enumerate(cfg["layers"]):
        for bi in range(nblocks):
            x = _clip_bottleneck(sd, f"{v}.layer{li + 1}.{bi}", x, 2 if (bi == 0 and li > 0) else 1)
    a = v + ".attnpool"
    b, c, h, w = x.shape
    t = x.flatten(2).permute(2, 0, 1)                                   # (HW) B C
    t = torch.cat([t.mean(dim=0, keepdim=True), t], dim=0) + sd[a + ".positional_embedding"][:, None, :]
    heads = cfg["heads"]
    d = c // heads
    q = F.linear(t[:1], sd[a + ".q_proj.weight"], sd[a + ".q_proj.bias"]) * d ** -0.5
    k = F.linear(t, sd[a + ".k_proj.weight"], sd[a + ".k_proj.bias"])
    vv = F.linear(t, sd[a + ".v_proj.weight"], sd[a + ".v_proj.bias"])
    q = q.reshape(1, b * heads, d).transpose(0, 1)                      # (B heads) 1 d
    k = k.reshape(-1, b * heads, d).transpose(0, 1)
    vv = vv.reshape(-1, b * heads, d).transpose(0, 1)
    att = torch.softmax(q @ k.transpose(1, 2), dim=-1) @ vv             # (B heads) 1 d
    o = att.transpose(0, 1).reshape(1, b, c)
    return F.linear(o, sd[a + ".c_proj.weight"], sd[a + ".c_proj.bias"])[0]


def clip_openai_text(sd, cfg, ids):
    """clip.model.CLIP.encode_text (TextEncoder, all_utils/utils.py:113-134): [B,77] int -> [B, embed_dim]."""
    x = sd["token_embedding.weight"][ids] + sd["positional_embedding"][None]
    n = x.shape[1]
    mask = torch.full((n, n), float("-inf")).triu_(1)
    heads = cfg["text_heads"]
    for i in range(cfg["text_layers"]):
        p = f"transformer.resblocks.{i}"
        h = F.layer_norm(x, (x.shape[-1],), sd[p + ".ln_1.weight"], sd[p + ".ln_1.bias"])
        qkv = F.linear(h, sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"])
        q, k, v = qkv.chunk(3, dim=-1)
        b, _, c = q.shape
        d = c // heads

        def sp(t):
            return t.reshape(b, n, heads, d).transpose(1, 2)
        att = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * d ** -0.5 + mask, dim=-1) @ sp(v)
        x = x + F.linear(att.transpose(1, 2).reshape(b, n, c), sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"])
        h = F.layer_norm(x, (x.shape[-1],), sd[p + ".ln_2.weight"], sd[p + ".ln_2.bias"])
        h = F.linear(h, sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"])
        x = x + F.linear(h * torch.sigmoid(1.702 * h), sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])
    x = F.layer_norm(x, (x.shape[-1],), sd["ln_final.weight"], sd["ln_final.bias"])
    return x[torch.arange(x.shape[0]), ids.argmax(dim=-1)] @ sd["text_projection"]


def clip_selector_logits(sd, cfg, pixels, ids):
    """CLIP_selector.forward (all_utils/utils.py:151-166): logit_scale.exp() * normalised image @ normalised text^T."""
    im = clip_rn50_visual(sd, cfg, pixels)
    tx = clip_openai_text(sd, cfg, ids)
    im = im / im.norm(dim=-1, keepdim=True)
    tx = tx / tx.norm(dim=-1, keepdim=True)
    return sd["logit_scale"].exp() * im @ tx.t()


# ---------------------------------------------------------------------------------------------------------------------
# WSDAN_CAL (fgvc/models/cal.py) on the fgvc ResNet features (fgvc/models/resnet.py)
# ---------------------------------------------------------------------------------------------------------------------
def _res_bottleneck(sd, pfx, x, stride):
    out = F.relu(bn(sd, pfx + ".bn1", F.conv2d(x, sd[pfx + ".conv1.weight"])))
    out = F.relu(bn(sd, pfx + ".bn2", F.conv2d(out, sd[pfx + ".conv2.weight"], stride=stride, padding=1)))
    out = bn(sd, pfx + ".bn3", F.conv2d(out, sd[pfx + ".conv3.weight"]))
    identity = x
    if pfx + ".downsample.0.weight" in sd:
        identity = bn(sd, pfx + ".downsample.1", F.conv2d(x, sd[pfx + ".downsample.0.weight"], stride=stride))
    return F.relu(out + identity)


def wsdan_cal_logits(sd, cfg, x, eps=1e-6):
    """WSDAN_CAL.forward in eval mode -> p [B, num_classes] (the first element of the returned tuple).  `features` is
    nn.Sequential(conv1, bn1, relu, maxpool, layer1..layer4): keys features.0 / .1 / .4 .. .7."""
    f = "features"
    x = F.relu(bn(sd, f + ".1", F.conv2d(x, sd[f + ".0.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    strides = (1, 2, 2, 1)          # fgvc ResNet(..., stride=1): layer4 keeps the 14 x 14 grid ("resnet with stride= 16")
    for li, nblocks in enumerate(cfg["layers"]):
        for bi in range(nblocks):
            x = _res_bottleneck(sd, f"{f}.{4 + li}.{bi}", x, strides[li] if bi == 0 else 1)
    att = F.relu(bn(sd, "attentions.bn", F.conv2d(x, sd["attentions.conv.weight"]), eps=1e-3))     # BasicConv2d
    b, c, h, w = x.shape
    fm = (torch.einsum("imjk,injk->imn", att, x) / float(h * w)).reshape(b, -1)                     # BAP, pool = GAP
    fm = torch.sign(fm) * torch.sqrt(torch.abs(fm) + eps)
    fm = F.normalize(fm, dim=-1)
    return F.linear(fm * 100.0, sd["fc.weight"])


def semantic_pass(logits):
    """get_semantic_filtering: the positive prompt (index 0) wins."""
    return logits.argmax(dim=-1) == 0


def confidence_pass(logits, correct_label, top_k=10):
    """all_utils/utils.py:363-364: `correct_label in logits.topk(k)[1]`."""
    k = min(top_k, logits.shape[-1])
    return bool((logits.topk(k)[1] == correct_label).any())

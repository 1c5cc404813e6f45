"""Device-side image pre-processing of the CLIP image front-ends: Pillow's antialiased bicubic resize (two integer
passes of `saspa_resample_u8`), centre crop folded into the coefficient tables, rescale + normalise
(`saspa_u8_to_act_norm`).

Mirrors what the reference's pipelines do on the CPU through `CLIPImageProcessor` (the `feature_extractor` of
`StableDiffusionSafetyChecker`: resize shortest edge to 224 with PIL bicubic, centre-crop 224, /255, normalise;
SURVEY 8a a7.9) and `BlipImageProcessor` (resize to 224 x 224 bicubic; a8).  The coefficient tables depend only on the
sizes: they are computed once per (in, out, crop) on the host in double precision exactly as Pillow's
`precompute_coeffs` / `normalize_coeffs_8bpc` do and cached on the device; all pixel arithmetic runs in the kernels."""
import ctypes as C
import math
from functools import lru_cache

import numpy as np
import torch

from . import _lib
from .ops import _check_dev, _dt, _ptr, _stream

PRECISION_BITS = 32 - 8 - 2
CLIP_IMAGE_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_IMAGE_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0)}       # Pillow's BICUBIC / BILINEAR (filter, support)


@lru_cache(maxsize=64)
def resample_tables(in_len, out_len, first=0, count=None, filt="bicubic"):
    """Pillow `precompute_coeffs(inSize, 0, inSize, outSize, filter)` + `normalize_coeffs_8bpc` for output samples
    [first, first + count): -> (bounds int32 [count, 2], coeffs int32 [count, ksize], ksize)."""
    count = out_len - first if count is None else count
    kernel, base_support = FILTERS[filt]
    scale = filterscale = in_len / out_len
    if filterscale < 1.0:
        filterscale = 1.0
    support = base_support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((count, 2), np.int32)
    coeffs = np.zeros((count, ksize), np.int32)
    ss = 1.0 / filterscale
    for r in range(count):
        xx = first + r
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_len:
            xmax = in_len
        xmax -= xmin
        w = [kernel((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)            # left-to-right double sum, like the C loop
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            coeffs[r, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[r] = (xmin, xmax)
    return bounds, coeffs, ksize


_DEV_TABLES = {}


def _tables_on(dev, in_len, out_len, first, count, filt="bicubic"):
    key = (str(dev), in_len, out_len, first, count, filt)
    if key not in _DEV_TABLES:
        b, c, k = resample_tables(in_len, out_len, first, count, filt)
        _DEV_TABLES[key] = (torch.from_numpy(b).to(dev), torch.from_numpy(c).to(dev), k)
    return _DEV_TABLES[key]


def resample_pass(src, axis, out_len, first=0, count=None, filt="bicubic"):
    """One pass over a device u8 [n,H,W,3] tensor along `axis` (1 = rows / vertical, 2 = columns / horizontal)."""
    _check_dev(src)
    if src.dtype != torch.uint8 or src.dim() != 4 or src.shape[3] != 3 or not src.is_contiguous():
        raise ValueError("resample_pass expects a contiguous u8 [n,H,W,3] tensor")
    lib = _lib.load()
    n, h, w, _ = src.shape
    count = out_len - first if count is None else count
    in_len = h if axis == 1 else w
    bounds, coeffs, ksize = _tables_on(src.device, in_len, out_len, first, count, filt)
    if axis == 1:
        dst = torch.empty((n, count, w, 3), device=src.device, dtype=torch.uint8)
        outer, inner = n, w * 3
    else:
        dst = torch.empty((n, h, count, 3), device=src.device, dtype=torch.uint8)
        outer, inner = n * h, 3
    _lib.check(lib.saspa_resample_u8(_ptr(src), _ptr(dst), outer, in_len, count, inner, _ptr(bounds), _ptr(coeffs), ksize,
                                     _stream()), "saspa_resample_u8")
    return dst


def resize_u8(src, out_h, out_w, crop=None, filt="bicubic"):
    """PIL.Image.resize((out_w, out_h), BICUBIC | BILINEAR) of every image of a device u8 [n,H,W,3] batch, optionally followed by
    the crop (top, left, height, width) -- horizontal pass first, then vertical, like Pillow; a pass whose size does
    not change is skipped, like Pillow."""
    n, h, w, _ = src.shape
    top, left, ch, cw = crop if crop is not None else (0, 0, out_h, out_w)
    x = src
    if out_w != w:
        x = resample_pass(x, 2, out_w, left, cw, filt)
    elif (left, cw) != (0, w):
        x = x[:, :, left:left + cw].contiguous()
    if out_h != h:
        x = resample_pass(x, 1, out_h, top, ch, filt)
    elif (top, ch) != (0, h):
        x = x[:, top:top + ch].contiguous()
    return x


def resize_bicubic_u8(src, out_h, out_w, crop=None):
    return resize_u8(src, out_h, out_w, crop, "bicubic")


def normalize_u8(src, dtype, mean=CLIP_IMAGE_MEAN, std=CLIP_IMAGE_STD):
    """u8 [n,H,W,3] -> [n,H,W,8] activations ((x/255) - mean) / std."""
    _check_dev(src)
    lib = _lib.load()
    n, h, w, _ = src.shape
    out = torch.empty((n, h, w, 8), device=src.device, dtype=dtype)
    _lib.check(lib.saspa_u8_to_act_norm(_dt(out), _ptr(src), _ptr(out), n * h * w, *[float(v) for v in mean],
                                        *[float(v) for v in std], _stream()), "saspa_u8_to_act_norm")
    return out


def clip_image_preprocess(src, dtype, size=224):
    """CLIPImageProcessor on a device u8 [n,H,W,3] batch: resize the shortest edge to `size` (long edge
    int(size * long / short)), centre-crop size x size, rescale, normalise -> [n,size,size,8]."""
    n, h, w, _ = src.shape
    if h <= w:
        oh, ow = size, int(size * w / h)
    else:
        oh, ow = int(size * h / w), size
    top, left = (oh - size) // 2, (ow - size) // 2
    x = resize_bicubic_u8(src.contiguous(), oh, ow, crop=(top, left, size, size))
    return normalize_u8(x, dtype)

"""Device-side image pre-processing of the CLIP image front-ends: Pillow's antialiased bicubic resize (two integer
passes of `saspa_resample_u8`), centre crop folded into the coefficient tables, rescale + normalise
(`saspa_u8_to_act_norm`).

Mirrors what the reference's pipelines do on the CPU through `CLIPImageProcessor` (the `feature_extractor` of
`StableDiffusionSafetyChecker`: resize shortest edge to 224 with PIL bicubic, centre-crop 224, /255, normalise;
SURVEY 8a a7.9) and `BlipImageProcessor` (resize to 224 x 224 bicubic; a8).  The coefficient tables depend only on the
sizes: they are computed once per (in, out, crop) on the host in double precision exactly as Pillow's
`precompute_coeffs` / `normalize_coeffs_8bpc` do and cached on the device; all pixel arithmetic runs in the kernels."""
import collections
import math
from functools import lru_cache

import numpy as np
import torch

from . import _lib
from . import ops
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


# ---------------------------------------------------------------------------------------------------------------------
# cv2.resize (8-bit RGB) as the reference's `resize_image` calls it (all_utils/utils.py:58-79): the tables
# are host set-up in OpenCV's own arithmetic (float32 source coordinates and weights, 11-bit fixed point, double-precision
# area cells); all pixel work runs in saspa_resize_taps_u8 / saspa_resize_area_u8.
# ---------------------------------------------------------------------------------------------------------------------
_COEF_SCALE = np.float32(2048.0)          # INTER_RESIZE_COEF_SCALE (11 bits)


def _lanczos4_weights(x):
    """OpenCV interpolateLanczos4(float x, float* coeffs)."""
    s45 = 0.70710678118654752440084436210485
    cs = ((1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1), (-s45, s45))
    c = np.zeros(8, np.float32)
    if x < np.finfo(np.float32).eps:
        c[3] = 1.0
        return c
    xd = float(x)
    y0 = -(xd + 3) * math.pi * 0.25
    s0, c0 = math.sin(y0), math.cos(y0)
    total = np.float32(0)
    for i in range(8):
        y = -(xd + 3 - i) * math.pi * 0.25
        c[i] = np.float32((cs[i][0] * s0 + cs[i][1] * c0) / (y * y))
        total = np.float32(total + c[i])
    return (c * np.float32(np.float32(1.0) / total)).astype(np.float32)


def _fix(weights):
    return np.clip(np.rint(weights.astype(np.float32) * _COEF_SCALE), -32768, 32767).astype(np.int16)     # saturate_cast<short>


@lru_cache(maxsize=256)
def cv_tap_tables(ssize, dsize, mode):
    """mode "lanczos4": first tap sx - 3 and 8 weights per destination sample; mode "area_linear": the bilinear taps of
    INTER_AREA's up-scaling branch (`area_mode` coordinates)."""
    scale = ssize / float(dsize)
    inv = dsize / float(ssize)
    nt = 8 if mode == "lanczos4" else 2
    ofs = np.zeros(dsize, np.int32)
    w = np.zeros((dsize, nt), np.int16)
    for d in range(dsize):
        if mode == "lanczos4":
            fx = np.float32((d + 0.5) * scale - 0.5)
            sx = int(math.floor(float(fx)))
            fx = np.float32(fx - np.float32(sx))
            ofs[d], w[d] = sx - 3, _fix(_lanczos4_weights(fx))
        else:
            sx = int(math.floor(d * scale))
            fx = np.float32((d + 1) - (sx + 1) * inv)
            fx = np.float32(0.0) if fx <= 0 else np.float32(fx - np.float32(math.floor(float(fx))))
            if sx < 0:
                fx, sx = np.float32(0.0), 0
            if sx >= ssize - 1:
                fx, sx = np.float32(0.0), ssize - 1
            ofs[d], w[d] = sx, _fix(np.array([np.float32(1.0) - fx, fx], np.float32))
    return ofs, w


@lru_cache(maxsize=256)
def cv_area_tables(ssize, dsize):
    """OpenCV computeResizeAreaTab as CSR: (start [dsize + 1], source index [nnz], float32 weight [nnz])."""
    scale = ssize / float(dsize)
    start, si, al = [0], [], []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            si.append(sx1 - 1)
            al.append(np.float32((sx1 - fsx1) / cell))
        for sx in range(sx1, sx2):
            si.append(sx)
            al.append(np.float32(1.0 / cell))
        if fsx2 - sx2 > 1e-3:
            si.append(sx2)
            al.append(np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell))
        start.append(len(si))
    return np.asarray(start, np.int32), np.asarray(si, np.int32), np.asarray(al, np.float32)


_CV_DEV = collections.OrderedDict()
_CV_DEV_MAX = 256          # (source size, target size, mode) table sets kept per process: real datasets have hundreds of sizes


def _cv_dev(dev, key, build):
    """Device copies of the host-built resize tables, LRU-bounded.  The upload goes through ops.h2d (pinned, non-blocking):
    a pageable .to(dev) is a synchronous copy that waits for everything queued on the stream -- the previous batch's
    whole step graph -- on every new source size."""
    k = (str(dev),) + key
    hit = _CV_DEV.get(k)
    if hit is not None:
        _CV_DEV.move_to_end(k)
        return hit
    val = tuple(ops.h2d(torch.from_numpy(np.ascontiguousarray(a)), dev) for a in build())
    _CV_DEV[k] = val
    while len(_CV_DEV) > _CV_DEV_MAX:
        _CV_DEV.popitem(last=False)
    return val


def cv_resize_u8(src, dh, dw, interpolation):
    """cv2.resize(src, (dw, dh), interpolation=INTER_LANCZOS4 | INTER_AREA) of a device u8 [n,H,W,3] batch."""
    _check_dev(src)
    if src.dtype != torch.uint8 or src.dim() != 4 or src.shape[3] != 3 or not src.is_contiguous():
        raise ValueError("cv_resize_u8 expects a contiguous u8 [n,H,W,3] tensor")
    n, h, w, _ = src.shape
    if (h, w) == (dh, dw):
        return src
    lib = _lib.load()
    dst = torch.empty((n, dh, dw, 3), device=src.device, dtype=torch.uint8)
    if interpolation == "lanczos4" or (interpolation == "area" and (w / float(dw) < 1 or h / float(dh) < 1)):
        mode = "lanczos4" if interpolation == "lanczos4" else "area_linear"
        xo, xw = _cv_dev(src.device, ("tap", w, dw, mode), lambda: cv_tap_tables(w, dw, mode))
        yo, yw = _cv_dev(src.device, ("tap", h, dh, mode), lambda: cv_tap_tables(h, dh, mode))
        _lib.check(lib.saspa_resize_taps_u8(_ptr(src), _ptr(dst), n, h, w, dh, dw, _ptr(xo), _ptr(xw), _ptr(yo), _ptr(yw),
                                            8 if mode == "lanczos4" else 2, 0 if mode == "lanczos4" else 1, _stream()),
                   "saspa_resize_taps_u8")
        return dst
    if interpolation != "area":
        raise ValueError(f"unsupported interpolation {interpolation!r}")
    sx, sy = w / float(dw), h / float(dh)
    isx, isy = int(round(sx)), int(round(sy))
    eps = np.finfo(np.float64).eps
    if abs(sx - isx) < eps and abs(sy - isy) < eps:                        # resizeAreaFast_
        _lib.check(lib.saspa_resize_area_u8(_ptr(src), _ptr(dst), n, h, w, dh, dw, None, None, None, None, None, None, isx, isy,
                                            _stream()), "saspa_resize_area_u8")
        return dst
    xs, xi, xa = _cv_dev(src.device, ("area", w, dw), lambda: cv_area_tables(w, dw))
    ys, yi, ya = _cv_dev(src.device, ("area", h, dh), lambda: cv_area_tables(h, dh))
    _lib.check(lib.saspa_resize_area_u8(_ptr(src), _ptr(dst), n, h, w, dh, dw, _ptr(xs), _ptr(xi), _ptr(xa), _ptr(ys), _ptr(yi),
                                        _ptr(ya), 0, 0, _stream()), "saspa_resize_area_u8")
    return dst

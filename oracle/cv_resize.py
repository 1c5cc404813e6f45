"""TEST INFRASTRUCTURE ONLY -- numpy restatement of ``cv2.resize`` for 8-bit 3-channel images in the two modes the reference's
``resize_image`` uses (all_utils/utils.py:58-79): ``INTER_LANCZOS4`` when up-scaling (k > 1), ``INTER_AREA`` when
down-scaling.

PARITY UNPINNED: opencv-python (pinned 4.8.0.74 by the reference's environment.yml:24) is not installable here and the
reference holds no image fixtures; this follows OpenCV's published ``modules/imgproc/src/resize.cpp`` (recalled):

* Lanczos4, 8U (``resizeGeneric_<HResizeLanczos4<uchar,int,short>, VResizeLanczos4<uchar,int,short, FixedPtCast<int,uchar,
  22>>>``): source coordinate ``fx = (float)((dx + 0.5) * scale - 0.5)``, ``sx = floor(fx)``, 8 taps ``sx-3 .. sx+4``
  clamped to the image (replicate), float weights from ``interpolateLanczos4`` (sin / cos recurrence over 45-degree steps,
  normalised to sum 1), converted to 11-bit fixed point with ``saturate_cast<short>(w * 2048)`` (round half to even); the
  horizontal pass keeps 32-bit integers, the vertical pass rounds once: ``(v + 2^21) >> 22``, saturated to 0..255.
* Area, 8U, non-integer scale (``resizeArea_<uchar, float>``): per destination cell a list of (source index, float weight)
  from ``computeResizeAreaTab`` (fractional first / last source pixel, ``1 / cellWidth`` in between, 1e-3 slack); rows are
  accumulated in float32 in table order (``buf += S * alpha`` horizontally, ``sum += beta * buf`` vertically, separate
  multiply and add roundings), the result is ``saturate_cast<uchar>`` = round half to even.  Integer scale factors take
  OpenCV's ``resizeAreaFast_`` (integer block mean).  When either scale factor is below 1 (a side that the /64 rounding
  pushes ABOVE the source size while k <= 1, e.g. 512x700 -> 512x704) OpenCV runs its bilinear code with the ``area_mode``
  source coordinates instead: :func:`resize_area_upscaling`.
"""
import math

import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def _cv_round_f32(v):
    """cvRound / saturate_cast from float: round half to even."""
    return np.rint(v)


def lanczos4_coeffs(x):
    """interpolateLanczos4(float x, float* coeffs)."""
    s45 = 0.70710678118654752440084436210485
    cs = ((1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1), (-s45, s45))
    x = np.float32(x)
    c = np.zeros(8, np.float32)
    if x < np.finfo(np.float32).eps:
        c[3] = 1.0
        return c
    y0 = -(float(x) + 3) * math.pi * 0.25
    s0, c0 = math.sin(y0), math.cos(y0)
    total = np.float32(0)
    for i in range(8):
        y = -(float(x) + 3 - i) * math.pi * 0.25
        c[i] = np.float32((cs[i][0] * s0 + cs[i][1] * c0) / (y * y))
        total = np.float32(total + c[i])
    inv = np.float32(np.float32(1.0) / total)
    return (c * inv).astype(np.float32)


def lanczos4_tables(ssize, dsize):
    """-> (first tap index sx-3 per destination sample [dsize], int16 weights [dsize, 8])."""
    scale = ssize / float(dsize)                     # 1 / inv_scale, double
    ofs = np.zeros(dsize, np.int64)
    w = np.zeros((dsize, 8), np.int64)
    for d in range(dsize):
        fx = np.float32((d + 0.5) * scale - 0.5)
        sx = int(math.floor(float(fx)))
        fx = np.float32(fx - np.float32(sx))
        cf = lanczos4_coeffs(fx)
        q = _cv_round_f32(cf * np.float32(INTER_RESIZE_COEF_SCALE))
        w[d] = np.clip(q, -32768, 32767).astype(np.int64)
        ofs[d] = sx - 3
    return ofs, w


def resize_lanczos4(img, dh, dw):
    h, w, _ = img.shape
    xo, xa = lanczos4_tables(w, dw)
    yo, ya = lanczos4_tables(h, dh)
    s = img.astype(np.int64)
    hor = np.zeros((h, dw, 3), np.int64)
    for k in range(8):
        idx = np.clip(xo + k, 0, w - 1)
        hor += s[:, idx, :] * xa[:, k][None, :, None]
    hor = ((hor + (1 << 31)) % (1 << 32)) - (1 << 31)            # int32 storage of the horizontal pass
    out = np.zeros((dh, dw, 3), np.int64)
    for k in range(8):
        idy = np.clip(yo + k, 0, h - 1)
        out += hor[idy] * ya[:, k][:, None, None]
    out = ((out + (1 << 31)) % (1 << 32)) - (1 << 31)
    out = (out + (1 << (2 * INTER_RESIZE_COEF_BITS - 1))) >> (2 * INTER_RESIZE_COEF_BITS)
    return np.clip(out, 0, 255).astype(np.uint8)


def area_tab(ssize, dsize, scale):
    """computeResizeAreaTab (cn = 1): list of (destination index, source index, float32 alpha) in table order."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area(img, dh, dw):
    h, w, _ = img.shape
    sx, sy = w / float(dw), h / float(dh)
    if (h, w) == (dh, dw):
        return img.copy()
    if sx < 1 or sy < 1:
        return resize_area_upscaling(img, dh, dw)
    isx, isy = int(round(sx)), int(round(sy))
    if abs(sx - isx) < np.finfo(np.float64).eps and abs(sy - isy) < np.finfo(np.float64).eps:
        # resizeAreaFast_: integer block mean
        area = isx * isy
        blk = img[:dh * isy, :dw * isx].reshape(dh, isy, dw, isx, 3).astype(np.int64).sum(axis=(1, 3))
        if isx == 2 and isy == 2:
            return ((blk + 2) >> 2).astype(np.uint8)
        return np.clip(np.rint(blk.astype(np.float32) * np.float32(1.0 / area)), 0, 255).astype(np.uint8)
    xt, yt = area_tab(w, dw, sx), area_tab(h, dh, sy)
    s = img.astype(np.float32)
    out = np.zeros((dh, dw, 3), np.uint8)
    acc = np.zeros((dw, 3), np.float32)
    prev = yt[0][0]
    for (dy, syi, beta) in yt:
        buf = np.zeros((dw, 3), np.float32)
        for (dxi, sxi, alpha) in xt:
            buf[dxi] = buf[dxi] + s[syi, sxi] * alpha             # float32 multiply, then float32 add
        if dy != prev:
            out[prev] = np.clip(_cv_round_f32(acc), 0, 255).astype(np.uint8)
            acc = beta * buf
            prev = dy
        else:
            acc = acc + beta * buf
    out[prev] = np.clip(_cv_round_f32(acc), 0, 255).astype(np.uint8)
    return out


def linear_area_tables(ssize, dsize):
    """INTER_AREA with a scale factor below 1 in either direction falls through to OpenCV's bilinear code with the
    `area_mode` source coordinates: sx = floor(dx * scale), fx = (dx + 1) - (sx + 1) * inv_scale, fx <= 0 ? 0 : fx - floor(fx);
    weights (1 - fx, fx) in 11-bit fixed point; taps beyond the last source sample collapse onto it."""
    scale = ssize / float(dsize)
    inv = dsize / float(ssize)
    ofs = np.zeros(dsize, np.int64)
    w = np.zeros((dsize, 2), np.int64)
    for d in range(dsize):
        sx = int(math.floor(d * scale))
        fx = np.float32((d + 1) - (sx + 1) * inv)
        fx = np.float32(0.0) if fx <= 0 else np.float32(fx - np.float32(math.floor(float(fx))))
        if sx < 0:
            fx, sx = np.float32(0.0), 0
        if sx >= ssize - 1:
            fx, sx = np.float32(0.0), ssize - 1
        cf = np.array([np.float32(1.0) - fx, fx], np.float32)
        w[d] = np.clip(_cv_round_f32(cf * np.float32(INTER_RESIZE_COEF_SCALE)), -32768, 32767).astype(np.int64)
        ofs[d] = sx
    return ofs, w


def resize_area_upscaling(img, dh, dw):
    """cv2.resize(..., INTER_AREA) when scale_x < 1 or scale_y < 1: HResizeLinear<uchar,int,short> + the 8-bit
    VResizeLinear specialisation ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 >> 2."""
    h, w, _ = img.shape
    xo, xa = linear_area_tables(w, dw)
    yo, ya = linear_area_tables(h, dh)
    s = img.astype(np.int64)
    hor = s[:, np.clip(xo, 0, w - 1), :] * xa[:, 0][None, :, None] + s[:, np.clip(xo + 1, 0, w - 1), :] * xa[:, 1][None, :, None]
    s0, s1 = hor[np.clip(yo, 0, h - 1)], hor[np.clip(yo + 1, 0, h - 1)]
    b0, b1 = ya[:, 0][:, None, None], ya[:, 1][:, None, None]
    out = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_image(input_image, smaller_side_res):
    """all_utils/utils.py:58-79 with the two cv2.resize modes restated above."""
    MAX_RES_SIZE = 1200000
    H, W, _ = input_image.shape
    H, W = float(H), float(W)
    k = float(smaller_side_res) / min(H, W)
    H *= k
    W *= k
    if H * W > MAX_RES_SIZE:
        k = np.sqrt(MAX_RES_SIZE / (H * W))
        H *= k
        W *= k
    H = int(np.round(H / 64.0)) * 64
    W = int(np.round(W / 64.0)) * 64
    return resize_lanczos4(input_image, H, W) if k > 1 else resize_area(input_image, H, W)

"""TEST INFRASTRUCTURE ONLY -- integer-exact numpy restatement of
``cv2.Canny(img_u8[H,W,C], low, high)`` as called by the reference
(all_utils/utils.py:81-85, aperture 3, L2gradient=False), plus the HWC3 /
3-channel replication around it (all_utils/utils.py:39-55, :87-99).

PARITY UNPINNED: opencv-python==4.8.0.74 (environment.yml:24) is not installed;
this follows the published algorithm of OpenCV's generic (non-IPP) path in
modules/imgproc/src/canny.cpp:

* Sobel 3x3 to int16 with BORDER_REPLICATE,
* L1 magnitude |dx|+|dy|; multi-channel input keeps, per pixel, the channel
  with the largest magnitude (first one on ties),
* non-maximum suppression with the fixed-point tangent test
  TG22 = round(tan(22.5deg) * 2^15) = 13573, asymmetric (>, >=) comparisons on
  the horizontal / vertical sectors and (>, >) on the diagonals, magnitudes
  outside the image = 0,
* hysteresis: pixels with mag > high seed; 8-connected pixels that passed NMS
  with low < mag <= high are added transitively,
* output 255 on edges, 0 elsewhere.
"""
import numpy as np

TG22 = 13573
CANNY_SHIFT = 15


def hwc3(x):
    """all_utils/utils.py:39-55"""
    assert x.dtype == np.uint8
    if x.ndim == 2:
        x = x[:, :, None]
    h, w, c = x.shape
    assert c in (1, 3, 4)
    if c == 3:
        return x
    if c == 1:
        return np.concatenate([x, x, x], axis=2)
    color = x[:, :, 0:3].astype(np.float32)
    alpha = x[:, :, 3:4].astype(np.float32) / 255.0
    y = color * alpha + 255.0 * (1.0 - alpha)
    return y.clip(0, 255).astype(np.uint8)


def sobel3_replicate(img):
    """img: u8 [H,W,C] -> (dx, dy) int32 [H,W,C] (values fit int16)."""
    p = np.pad(img.astype(np.int32), ((1, 1), (1, 1), (0, 0)), mode="edge")
    tl, tc, tr = p[:-2, :-2], p[:-2, 1:-1], p[:-2, 2:]
    ml, mr = p[1:-1, :-2], p[1:-1, 2:]
    bl, bc, br = p[2:, :-2], p[2:, 1:-1], p[2:, 2:]
    dx = (tr + 2 * mr + br) - (tl + 2 * ml + bl)
    dy = (bl + 2 * bc + br) - (tl + 2 * tc + tr)
    return dx, dy


def canny_nms_map(img, low, high):
    """Returns the OpenCV 'map' before hysteresis: 2 strong, 0 weak candidate, 1 none."""
    if img.ndim == 2:
        img = img[:, :, None]
    if low > high:
        low, high = high, low
    low = int(np.floor(low))
    high = int(np.floor(high))
    dx, dy = sobel3_replicate(img)
    mag = np.abs(dx) + np.abs(dy)
    idx = np.argmax(mag, axis=2)          # first max on ties == OpenCV's strict '>' scan
    dx = np.take_along_axis(dx, idx[:, :, None], 2)[:, :, 0]
    dy = np.take_along_axis(dy, idx[:, :, None], 2)[:, :, 0]
    mag = np.take_along_axis(mag, idx[:, :, None], 2)[:, :, 0]

    m = np.pad(mag, 1)                    # zero border
    c = m[1:-1, 1:-1]
    left, right = m[1:-1, :-2], m[1:-1, 2:]
    up, down = m[:-2, 1:-1], m[2:, 1:-1]
    ul, ur = m[:-2, :-2], m[:-2, 2:]
    dl, dr = m[2:, :-2], m[2:, 2:]

    x = np.abs(dx).astype(np.int64)
    y = np.abs(dy).astype(np.int64) << CANNY_SHIFT
    tg22x = x * TG22
    tg67x = tg22x + (x << (CANNY_SHIFT + 1))
    horiz = y < tg22x
    vert = (~horiz) & (y > tg67x)
    diag = (~horiz) & (~vert)
    neg = (dx ^ dy) < 0                   # s = -1
    # s=+1: prev row j-1 (ul), next row j+1 (dr); s=-1: prev row j+1 (ur), next row j-1 (dl)
    d_prev = np.where(neg, ur, ul)
    d_next = np.where(neg, dl, dr)
    keep = (horiz & (c > left) & (c >= right)) | (vert & (c > up) & (c >= down)) | (diag & (c > d_prev) & (c > d_next))
    keep &= c > low
    out = np.ones(c.shape, np.uint8)
    out[keep & (c > high)] = 2
    out[keep & (c <= high)] = 0
    return out


def hysteresis(cmap):
    """Transitive 8-connected growth of strong (2) into weak (0) pixels."""
    h, w = cmap.shape
    m = np.pad(cmap, 1, constant_values=1)
    stack = list(zip(*np.nonzero(m == 2)))
    while stack:
        i, j = stack.pop()
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                if m[i + di, j + dj] == 0:
                    m[i + di, j + dj] = 2
                    stack.append((i + di, j + dj))
    return m[1:-1, 1:-1]


def canny(img, low, high):
    """u8 [H,W] or [H,W,C] -> u8 [H,W] in {0,255}."""
    m = hysteresis(canny_nms_map(img, low, high))
    return np.where(m == 2, 255, 0).astype(np.uint8)


def generate_canny_array(img_u8, low, high):
    """generate_canny minus the resize (identity when the input is already at
    ``image_resolution`` with /64 sides): HWC3 -> Canny -> HWC3.  all_utils/utils.py:87-109"""
    return hwc3(canny(hwc3(img_u8), low, high))

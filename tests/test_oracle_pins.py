"""Independent cross-checks of the two integer oracles that nothing in the reference can pin (no cv2 here, no image fixtures
in the reference: SURVEY 8(c)).  They do not change the parity grade -- the oracles still restate OpenCV from its published
source -- but they remove the places where a recall error could hide behind bit-exact HIP-vs-oracle agreement:

* `oracle/canny.py: sobel3_replicate` against `scipy.ndimage.sobel(mode="nearest")` (an independent implementation of the
  same 3x3 derivative with replicated borders; cv2.Canny -> cv::Sobel(..., BORDER_REPLICATE), all_utils/utils.py:81-109);
* `oracle/cv_resize.py` (all_utils/utils.py:58-79): the INTER_AREA integer-scale path against exact block means in rational
  arithmetic, the Lanczos4 weights against the closed-form kernel sinc(x) sinc(x / 4) normalised over its 8 taps, and the
  non-integer area table against the exact overlap lengths of the destination cell with the source pixels.
"""
import math
from fractions import Fraction

import numpy as np
import pytest

from oracle import canny as OC
from oracle import cv_resize as CR

scipy_ndimage = pytest.importorskip("scipy.ndimage")


@pytest.mark.parametrize("shape", [(17, 23, 3), (64, 64, 3), (5, 3, 1), (1, 9, 3), (9, 1, 3)])
def test_sobel_replicate_equals_scipy_nearest(shape):
    img = np.random.RandomState(sum(shape)).randint(0, 256, shape).astype(np.uint8)
    dx, dy = OC.sobel3_replicate(img)
    x = img.astype(np.int32)
    for c in range(shape[2]):
        # scipy's sobel = derivative [-1, 0, 1] along `axis`, smoothing [1, 2, 1] along the other: cv::Sobel's 3x3 kernels
        assert np.array_equal(dx[..., c], scipy_ndimage.sobel(x[..., c], axis=1, mode="nearest"))
        assert np.array_equal(dy[..., c], scipy_ndimage.sobel(x[..., c], axis=0, mode="nearest"))


def test_sobel_known_answers():
    """A vertical step edge 0 | 255: dx = 4 * 255 on the two columns beside the step (1 + 2 + 1 rows), dy = 0; borders
    replicate, so the outer columns see no gradient."""
    img = np.zeros((5, 6, 1), np.uint8)
    img[:, 3:] = 255
    dx, dy = OC.sobel3_replicate(img)
    assert (dy == 0).all()
    assert (dx[:, 2, 0] == 1020).all() and (dx[:, 3, 0] == 1020).all()
    assert (dx[:, [0, 1, 4, 5], 0] == 0).all()


@pytest.mark.parametrize("k", [2, 3, 4])
def test_area_integer_scale_is_the_rounded_block_mean(k):
    """resizeAreaFast_: every destination pixel is the mean of its k x k source block; round half to even for k != 2
    (saturate_cast of the float product), (sum + 2) >> 2 for the 2 x 2 case -- checked against exact rationals.  The
    float32 product `sum * (1 / area)` may differ from the exact mean only when the exact mean sits within float rounding of a
    half: those pixels are excluded (none at k = 2 / 4, where 1 / area is a power of two)."""
    rs = np.random.RandomState(k)
    img = rs.randint(0, 256, (6 * k, 8 * k, 3)).astype(np.uint8)
    got = CR.resize_area(img, 6, 8)
    blk = img.reshape(6, k, 8, k, 3).astype(np.int64).sum(axis=(1, 3))
    n_checked = 0
    for y in range(6):
        for x in range(8):
            for c in range(3):
                mean = Fraction(int(blk[y, x, c]), k * k)
                if k == 2:
                    want = (int(blk[y, x, c]) + 2) >> 2
                else:
                    frac = mean - math.floor(mean)
                    if abs(frac - Fraction(1, 2)) < Fraction(1, 1000) and k == 3:
                        continue
                    want = math.floor(mean) + (1 if frac > Fraction(1, 2) else 0)
                    if frac == Fraction(1, 2):
                        want = math.floor(mean) + (math.floor(mean) & 1)          # half to even
                assert int(got[y, x, c]) == want, (y, x, c, mean)
                n_checked += 1
    assert n_checked > 100


def _lanczos_closed_form(fx):
    """8 taps at distances (fx + 3 - i), i = 0..7 from the sample: L(x) = sinc(x) sinc(x / 4), |x| < 4, normalised to sum 1."""
    def sinc(v):
        return 1.0 if v == 0 else math.sin(math.pi * v) / (math.pi * v)
    w = [sinc(fx + 3 - i) * sinc((fx + 3 - i) / 4.0) for i in range(8)]
    s = sum(w)
    return [v / s for v in w]


@pytest.mark.parametrize("fx", [0.0, 0.125, 0.25, 1.0 / 3, 0.5, 0.625, 0.75, 0.9, 0.999])
def test_lanczos4_weights_equal_the_closed_form_kernel(fx):
    """interpolateLanczos4 evaluates sin(pi x) sin(pi x / 4) / x^2 through a 45-degree recurrence; up to the normalisation that is
    the Lanczos kernel with a = 4.  The oracle's float32 weights must agree with the closed form to float precision, and the
    11-bit fixed-point table built from them to one unit (it is what the 8-bit path multiplies with)."""
    got = CR.lanczos4_coeffs(np.float32(fx)).astype(np.float64)
    want = np.array(_lanczos_closed_form(float(np.float32(fx))))
    assert abs(got.sum() - 1.0) < 1e-6
    assert np.abs(got - want).max() < 2e-6, (got, want)
    assert np.abs(np.rint(got * 2048) - np.rint(want * 2048)).max() <= 1


def test_lanczos4_tables_source_coordinates():
    """Destination sample d reads the source at (d + 0.5) * scale - 0.5 (pixel-centre alignment): at an exact 2 x up-scale the
    coordinates are -0.25, 0.25, 0.75, 1.25 ... -> first tap floor(.) - 3 and phases 0.75 / 0.25 alternating."""
    ofs, w = CR.lanczos4_tables(8, 16)
    assert list(ofs[:4]) == [-1 - 3, 0 - 3, 0 - 3, 1 - 3]
    assert np.array_equal(w[1], w[3]) and np.array_equal(w[2], w[4]) and not np.array_equal(w[1], w[2])
    assert (w.sum(axis=1) >= 2046).all() and (w.sum(axis=1) <= 2050).all()          # 11-bit weights sum to ~2048
    # mirror symmetry of the kernel: the weights at phase 0.25 are those at phase 0.75 reversed
    assert np.abs(w[1] - w[2][::-1]).max() <= 1


@pytest.mark.parametrize("ssize,dsize", [(700, 512), (683, 512), (1024, 704), (333, 64), (65, 64)])
def test_area_table_equals_exact_overlaps(ssize, dsize):
    """computeResizeAreaTab: the weight of source pixel s in destination cell d is |[d * scale, (d + 1) * scale) n [s, s + 1)| / cell
    width -- checked against the overlap computed in rational arithmetic (entries below OpenCV's 1e-3 slack may be dropped)."""
    scale = ssize / float(dsize)
    tab = CR.area_tab(ssize, dsize, scale)
    fs = Fraction(ssize, dsize)
    got = {}
    for d, s, a in tab:
        assert (d, s) not in got
        got[(d, s)] = float(a)
    for d in range(dsize):
        lo, hi = d * fs, min((d + 1) * fs, Fraction(ssize))
        cell = hi - lo
        tot = 0.0
        for s in range(math.floor(lo), min(math.ceil(hi), ssize)):
            ov = min(hi, Fraction(s + 1)) - max(lo, Fraction(s))
            want = float(ov / cell)
            if (d, s) in got:
                assert abs(got[(d, s)] - want) < 1e-5, (d, s, got[(d, s)], want)
                tot += got[(d, s)]
            else:
                assert float(ov) <= 1e-3 + 1e-9, (d, s, float(ov))               # dropped only inside the slack
        assert abs(tot - 1.0) < 2e-3
    # order: destination-major, sources ascending (the accumulation order the float32 result depends on)
    assert [t[:2] for t in tab] == sorted(t[:2] for t in tab)


def test_area_non_integer_scale_close_to_exact_area_mean():
    """resizeArea_ accumulates in float32: the 8-bit result is within one level of the exact area-weighted mean."""
    rs = np.random.RandomState(5)
    img = rs.randint(0, 256, (45, 70, 3)).astype(np.uint8)
    dh, dw = 32, 48
    got = CR.resize_area(img, dh, dw).astype(np.float64)
    sy, sx = Fraction(45, dh), Fraction(70, dw)

    def overlaps(n_src, d, sc):
        lo, hi = d * sc, (d + 1) * sc
        return [(s, float(min(hi, Fraction(s + 1)) - max(lo, Fraction(s))) / float(sc)) for s in range(math.floor(lo), min(math.ceil(hi), n_src))]
    x = img.astype(np.float64)
    want = np.zeros((dh, dw, 3))
    for y in range(dh):
        for (syi, wy) in overlaps(45, y, sy):
            for xx in range(dw):
                for (sxi, wx) in overlaps(70, xx, sx):
                    want[y, xx] += wy * wx * x[syi, sxi]
    assert np.abs(got - want).max() <= 1.0 + 1e-6
    assert np.abs(got - np.rint(want)).mean() < 0.05

"""HED control images: the annotator the reference builds for CONTROLNET = "hed"
(`HEDdetector.from_pretrained('lllyasviel/ControlNet')`, run_aug/run_aug.py:311-312) and calls once per source image
(`hed_detector(orig_img)`, :438-439), as a launch sequence of the gfx950 kernels.

Network (controlnet_aux `ControlNetHED_Apache2`, checkpoint `ControlNetHED.pth`): x - norm, five blocks of 2/2/3/3/3
3x3 conv + ReLU (ReLU rides in the GEMM epilogue) with a 2x2 max-pool in front of blocks 2-5, a 1x1 side output per block;
head (`HEDdetector.__call__` defaults): the side outputs bilinearly resized (cv2 INTER_LINEAR, float32) to the image,
averaged, sigmoid, x255, truncated to u8, three identical channels -- one kernel, `saspa_hed_fuse`.

MI355X-first: a whole batch of source images per call (the reference runs one image at a time on the host's torch),
exact-fp32 MFMA convs (80 GFLOP per 512x512 image: 0.1 % of generating it; an annotator map thresholded by truncation
should not move with bf16 rounding), input u8 -> (x - norm) and the head on the device.  The detector's two internal
`resize_image(.., 512)` calls keep run_aug's images as they are (already 512 on the smaller side, multiples of 64);
other sizes are refused rather than silently resampled differently.  No CPU / eager arithmetic; fails without the .so."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib, imageproc, ops
from . import weights as W
from .config import HED


def _linear_tables(ssize, dsize):
    """cv2.resize INTER_LINEAR, float32 source (OpenCV resize.cpp): fx = (float)((d + 0.5) * scale - 0.5), sx = floor(fx),
    fx -= sx; columns: sx < 0 -> (0, f = 0), sx >= ssize - 1 -> (ssize - 1, f = 0); rows keep f and clamp the two indices."""
    scale = 1.0 / (dsize / float(ssize))
    xo = np.zeros(dsize, np.int32)
    xw = np.zeros((dsize, 2), np.float32)
    yo = np.zeros((dsize, 2), np.int32)
    yw = np.zeros((dsize, 2), np.float32)
    for d in range(dsize):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(float(f)))
        f = np.float32(f - np.float32(s))
        yo[d] = (min(max(s, 0), ssize - 1), min(max(s + 1, 0), ssize - 1))
        yw[d] = (np.float32(1.0) - f, f)
        if s < 0:
            f, s = np.float32(0.0), 0
        if s >= ssize - 1:
            f, s = np.float32(0.0), ssize - 1
        xo[d] = s
        xw[d] = (np.float32(1.0) - f, f)
    return xo, xw, yo, yw


class HEDdetector:
    """Call forms: `det(pil_image) -> PIL.Image` (the reference's), `det.detect_batch(u8 [n,H,W,3] device tensor) -> u8
    [n,H,W,3]` (run_aug's batched path)."""

    def __init__(self, sd, cfg=HED, device="cuda:0", dtype=torch.float32):
        self.cfg, self.dev, self.dtype = cfg, torch.device(device), dtype
        self.norm = [float(v) for v in sd["norm"].reshape(-1)]
        self.p = {}
        for i, (c, n) in enumerate(cfg["blocks"]):
            for j in range(n):
                self._pack(f"block{i + 1}.convs.{j}", sd)
            self._pack(f"block{i + 1}.projection", sd)
        self._tables = {}

    @classmethod
    def from_pretrained(cls, path, filename="ControlNetHED.pth", device="cuda:0"):
        """`path`: a local directory holding the annotator checkpoint (there is no hub access)."""
        import os
        f = os.path.join(path, filename)
        if not os.path.exists(f):
            raise FileNotFoundError(f"{f}: the HED annotator checkpoint must be on local disk")
        return cls(torch.load(f, map_location="cpu", weights_only=True), HED, device)

    def _pack(self, name, sd):
        w = sd[name + ".weight"].float()
        kh = w.shape[2]
        pk = W.pack_conv(w)
        chunk = W.chunk_major_ok(kh, kh, W.round8(w.shape[1]), 0, self.dtype)
        if chunk:
            pk = W.to_chunk_major(pk, kh * kh, self.dtype)
        t = pk.to(self.dev, self.dtype)
        t.saspa_korder = 1 if chunk else 0
        self.p[name + ".w"], self.p[name + ".b"] = t, sd[name + ".bias"].float().to(self.dev).contiguous()

    def side_outputs(self, img_u8):
        """u8 [n,H,W,3] on the device -> five fp32 [n,H>>k,W>>k,8] tensors (channel 0 live)."""
        p = self.p
        # x - norm[c] = ((x / 255) - norm[c] / 255) / (1 / 255)
        h = imageproc.normalize_u8(img_u8.contiguous(), self.dtype, [v / 255.0 for v in self.norm], (1 / 255.0,) * 3)
        outs = []
        for i, (_, n) in enumerate(self.cfg["blocks"]):
            if i > 0:
                h = ops.pool2d(h, 2, mode="max")
            for j in range(n):
                nm = f"block{i + 1}.convs.{j}"
                h = ops.conv(h, p[nm + ".w"], p[nm + ".b"], kh=3, kw=3, pad=1, act=ops.ACT_RELU)
            nm = f"block{i + 1}.projection"
            outs.append(ops.conv(h, p[nm + ".w"], p[nm + ".b"]).float())
        return outs

    def _tabs(self, src_hw, dst_hw):
        key = (src_hw, dst_hw)
        if key not in self._tables:
            xo, xw, _, _ = _linear_tables(src_hw[1], dst_hw[1])
            _, _, yo, yw = _linear_tables(src_hw[0], dst_hw[0])
            self._tables[key] = tuple(torch.from_numpy(np.ascontiguousarray(t)).to(self.dev) for t in (xo, xw, yo, yw))
        return self._tables[key]

    def detect_batch(self, img_u8):
        if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[3] != 3 or not img_u8.is_cuda:
            raise ValueError("HED input must be a device u8 [n,H,W,3] batch")
        n, hh, ww, _ = img_u8.shape
        if hh % 64 or ww % 64 or min(hh, ww) != 512:
            raise ValueError(f"HED input {hh}x{ww}: run_aug hands over images its resize_image brought to 512 on the smaller "
                             f"side and multiples of 64; the detector's own resize to other sizes is not built")
        outs = self.side_outputs(img_u8)
        dst = torch.empty((n, hh, ww, 3), device=self.dev, dtype=torch.uint8)
        q = _lib.HedFuseParams()
        q.nmaps, q.n, q.H, q.W = len(outs), n, hh, ww
        keep = []
        for k, o in enumerate(outs):
            mh, mw = o.shape[1], o.shape[2]
            xo, xw, yo, yw = self._tabs((mh, mw), (hh, ww))
            keep.append((o, xo, xw, yo, yw))
            q.map[k], q.mh[k], q.mw[k], q.ld[k] = o.data_ptr(), mh, mw, o.shape[3]
            q.xofs[k], q.xw[k], q.yofs[k], q.yw[k] = xo.data_ptr(), xw.data_ptr(), yo.data_ptr(), yw.data_ptr()
        q.dst = dst.data_ptr()
        _lib.check(_lib.load().saspa_hed_fuse(C.byref(q), ops._stream()), "saspa_hed_fuse")
        return dst

    def __call__(self, image):
        """The reference's call form: PIL image in, PIL image (RGB edge map) out."""
        from PIL import Image
        arr = np.asarray(image.convert("RGB"))
        out = self.detect_batch(torch.from_numpy(arr.copy())[None].to(self.dev))
        return Image.fromarray(out[0].cpu().numpy())

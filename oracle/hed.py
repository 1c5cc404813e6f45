"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the HED annotator the reference uses for CONTROLNET = "hed"
(run_aug/run_aug.py:20, :311-312 `HEDdetector.from_pretrained('lllyasviel/ControlNet')`, :438-439 `hed_detector(orig_img)`).

PARITY UNPINNED: `controlnet_aux` (pinned 0.0.5 by the reference's environment.yml:23) is a third-party package that is
neither in /root/reference nor installed here, and OpenCV is absent too.  This follows the package's published
`controlnet_aux/hed/__init__.py` (recalled): network `ControlNetHED_Apache2` (x - norm; five `DoubleConvBlock`s = 2/2/3/3/3
3x3 conv + ReLU with a 2x2 max-pool in front of blocks 2-5; a 1x1 `projection` per block), and `HEDdetector.__call__`
with its defaults (detect_resolution = image_resolution = 512, safe = False, scribble = False, output_type = "pil"):

    input_image = HWC3(np.array(pil)); input_image = resize_image(input_image, 512)        # identity for run_aug's inputs
    edges = net(float image, NCHW)                                                            # five [1,1,H>>k,W>>k] maps
    edges = [cv2.resize(e, (W, H), interpolation=cv2.INTER_LINEAR) for e in edges]            # float32 bilinear
    edge = 1 / (1 + np.exp(-np.mean(np.stack(edges, 2), 2).astype(np.float64)))
    edge = (edge * 255.0).clip(0, 255).astype(np.uint8); detected_map = HWC3(edge)
    detected_map = cv2.resize(detected_map, (W', H'), INTER_LINEAR) with (H', W') = resize_image(input_image, 512).shape   # identity

run_aug hands over images `resize_image` already brought to 512 on the smaller side and multiples of 64 (all_utils/
utils.py:58-79), so both `resize_image` calls inside the detector keep the size (k = 1 -> INTER_AREA at equal size = copy)."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def hed_network(sd, cfg, x):
    """x: [n,3,H,W] float (0..255 RGB) -> list of five side outputs [n,1,H>>k,W>>k]."""
    h = x - sd["norm"].float()
    outs = []
    for i, (_, n) in enumerate(cfg["blocks"]):
        if i > 0:
            h = F.max_pool2d(h, kernel_size=(2, 2), stride=(2, 2))
        for j in range(n):
            h = F.relu(F.conv2d(h, sd[f"block{i + 1}.convs.{j}.weight"].float(), sd[f"block{i + 1}.convs.{j}.bias"].float(), padding=1))
        outs.append(F.conv2d(h, sd[f"block{i + 1}.projection.weight"].float(), sd[f"block{i + 1}.projection.bias"].float()))
    return outs


def linear_tables_f32(ssize, dsize):
    """cv2.resize INTER_LINEAR coordinate / weight tables for a float32 image (resize.cpp, non-area mode):
    fx = (float)((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx; sx < 0 -> (0, 0); sx >= ssize - 1 -> (ssize - 1, 0)."""
    scale = 1.0 / (dsize / float(ssize))
    ofs = np.zeros(dsize, np.int64)
    w = np.zeros((dsize, 2), np.float32)
    for d in range(dsize):
        fx = np.float32((d + 0.5) * scale - 0.5)
        sx = int(math.floor(float(fx)))
        fx = np.float32(fx - np.float32(sx))
        if sx < 0:
            fx, sx = np.float32(0.0), 0
        if sx >= ssize - 1:
            fx, sx = np.float32(0.0), ssize - 1
        ofs[d] = sx
        w[d] = (np.float32(1.0) - fx, fx)
    return ofs, w


def linear_tables_rows_f32(ssize, dsize):
    """Vertical pass: the weight is kept, the two source rows are clamped into the image instead."""
    scale = 1.0 / (dsize / float(ssize))
    r0 = np.zeros(dsize, np.int64)
    r1 = np.zeros(dsize, np.int64)
    w = np.zeros((dsize, 2), np.float32)
    for d in range(dsize):
        fy = np.float32((d + 0.5) * scale - 0.5)
        sy = int(math.floor(float(fy)))
        fy = np.float32(fy - np.float32(sy))
        r0[d], r1[d] = min(max(sy, 0), ssize - 1), min(max(sy + 1, 0), ssize - 1)
        w[d] = (np.float32(1.0) - fy, fy)
    return r0, r1, w


def resize_linear_f32(img, dh, dw):
    """cv2.resize(img[h,w] float32, (dw, dh), INTER_LINEAR): separate float32 multiply / add roundings."""
    h, w_ = img.shape
    if (h, w_) == (dh, dw):
        return img.copy()
    xo, xa = linear_tables_f32(w_, dw)
    r0, r1, ya = linear_tables_rows_f32(h, dh)
    s = img.astype(np.float32)
    x1 = np.minimum(xo + 1, w_ - 1)
    hor = (s[:, xo] * xa[:, 0][None, :]).astype(np.float32) + (s[:, x1] * xa[:, 1][None, :]).astype(np.float32)
    hor = hor.astype(np.float32)
    out = (hor[r0] * ya[:, 0][:, None]).astype(np.float32) + (hor[r1] * ya[:, 1][:, None]).astype(np.float32)
    return out.astype(np.float32)


def hed_detect(sd, cfg, img_u8):
    """img_u8: [H,W,3] RGB uint8 (H, W multiples of 64, smaller side = the detect resolution) -> [H,W,3] uint8 edge map."""
    hh, ww, _ = img_u8.shape
    x = torch.from_numpy(img_u8.copy()).float().permute(2, 0, 1)[None]
    with torch.no_grad():
        edges = [e[0, 0].numpy().astype(np.float32) for e in hed_network(sd, cfg, x)]
    edges = [resize_linear_f32(e, hh, ww) for e in edges]
    acc = edges[0]
    for e in edges[1:]:
        acc = (acc + e).astype(np.float32)             # np.mean over a length-5 float32 axis: sequential float32 adds
    mean = (acc / np.float32(len(edges))).astype(np.float32)
    edge = 1.0 / (1.0 + np.exp(-mean.astype(np.float64)))
    edge = (edge * 255.0).clip(0, 255).astype(np.uint8)
    return np.stack([edge] * 3, axis=2)

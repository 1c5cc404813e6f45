"""Synthetic inputs of the bench / parity workloads: there are no datasets,
checkpoints or tokenizer vocabularies on the build or GPU boxes, so source images, prompts
and weights of the reference's shapes are generated from fixed seeds."""
import numpy as np


def synthetic_image(h, w, seed):
    """u8 [h, w, 3]: random filled rectangles / ellipses on a smooth gradient (gives Canny
    maps with a few percent edge pixels)."""
    rng = np.random.RandomState(1234 + seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        a, b = rng.uniform(-1, 1, 2)
        img[:, :, c] = 128 + 60 * (a * yy / max(h - 1, 1) + b * xx / max(w - 1, 1))
    n_shapes = 6 + rng.randint(0, 8)
    for _ in range(n_shapes):
        color = rng.randint(0, 256, 3).astype(np.float32)
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        ry, rx = rng.uniform(0.05, 0.3) * h, rng.uniform(0.05, 0.3) * w
        if rng.rand() < 0.5:
            mask = (np.abs(yy - cy) < ry) & (np.abs(xx - cx) < rx)
        else:
            mask = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
        img[mask] = color
    img += rng.normal(0, 2.0, img.shape)
    return np.clip(np.round(img), 0, 255).astype(np.uint8)


def synthetic_prompt_ids(n_prompts, seed=1, vocab=49408, max_len=77):
    """int64 [n, 77]: BOS + n random word ids + EOS padding, CLIP convention
    (BOS = vocab-2, EOS = PAD = vocab-1)."""
    rng = np.random.RandomState(seed)
    bos, eos = vocab - 2, vocab - 1
    ids = np.full((n_prompts, max_len), eos, np.int64)
    ids[:, 0] = bos
    for i in range(n_prompts):
        n = rng.randint(8, 41)
        ids[i, 1:1 + n] = rng.randint(0, vocab - 2, n)
    return ids


def negative_prompt_ids(vocab=49408, max_len=77, n=60, seed=7):
    rng = np.random.RandomState(seed)
    ids = np.full((1, max_len), vocab - 1, np.int64)
    ids[0, 0] = vocab - 2
    ids[0, 1:1 + n] = rng.randint(0, vocab - 2, n)
    return ids

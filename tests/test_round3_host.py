"""Round-3 host-side fixes (no GPU): the filter stage refuses to run on random weights unless asked to, fails BEFORE the
generation loop, decodes augmentations per batch instead of all at once, and the resize-table cache is bounded."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import saspa_aug_amd  # noqa: F401,E402
from saspa_aug_amd import dataset_utils, filters, imageproc  # noqa: E402
from saspa_aug_amd import run_aug as R  # noqa: E402


def _ds(tmp_path):
    return dataset_utils.SyntheticUtils(root_path=str(tmp_path / "data"), n_images=4, sizes=((64, 64),), print_func=lambda *a: None)


def test_filter_checkpoints_are_required(tmp_path, monkeypatch):
    """all_utils/dataset_utils.py:92 asserts exactly one baseline checkpoint and all_utils/utils.py:253 loads the real CLIP:
    a filter flag with no checkpoint behind it is an error, not a silent fall-back to random models."""
    ds = _ds(tmp_path)
    monkeypatch.delenv("SASPA_SYNTHETIC_FILTERS", raising=False)
    with pytest.raises(FileNotFoundError, match="semantic filter"):
        filters.filter_checkpoints(ds, None, semantic=True, confidence=False)
    with pytest.raises(FileNotFoundError, match="model-confidence filter"):
        filters.filter_checkpoints(ds, str(tmp_path / "w"), semantic=False, confidence=True)
    # present checkpoints are found; two baseline checkpoints are ambiguous (the reference's assert)
    w = tmp_path / "w"
    (w / "clip").mkdir(parents=True)
    (w / "clip" / "RN50.pt").write_bytes(b"x")
    (w / "checkpoints" / "synthetic").mkdir(parents=True)
    (w / "checkpoints" / "synthetic" / "a.pth").write_bytes(b"x")
    rn, cp = filters.filter_checkpoints(ds, str(w), True, True)
    assert rn.endswith("clip/RN50.pt") and cp.endswith("a.pth")
    (w / "checkpoints" / "synthetic" / "b.pth").write_bytes(b"x")
    with pytest.raises(FileNotFoundError, match="Expected 1"):
        filters.filter_checkpoints(ds, str(w), True, True)
    # explicit opt-in: nothing is required
    monkeypatch.setenv("SASPA_SYNTHETIC_FILTERS", "1")
    assert filters.filter_checkpoints(ds, None, True, True) == (None, None)


def test_main_refuses_filter_flags_without_checkpoints_before_generating(tmp_path, monkeypatch):
    """The check runs before the pipeline is built or a single image is generated."""
    monkeypatch.delenv("SASPA_SYNTHETIC_FILTERS", raising=False)
    ds = _ds(tmp_path)
    s = R.Settings(DATASET="synthetic", NUM_PER_IMAGE=1, RESOLUTION=64, USE_ARTISTIC_PROMPTS=False, PROMPT_WITH_SUB_CLASS=False,
                   SEMANTIC_FILTERING=1, MODEL_CONFIDENCE_BASED_FILTERING=1)

    class Boom:
        def __getattr__(self, name):
            raise AssertionError("the batch generator must not be touched")

    with pytest.raises(FileNotFoundError):
        R.main(s, ds_utils=ds, batch_generator=Boom())


def test_apply_filters_streams_batches(tmp_path, monkeypatch):
    """Pixels are decoded per batch (a prefetch of one batch), never the whole dataset: count what is alive at once."""
    ds = _ds(tmp_path)
    folder = tmp_path / "aug"
    folder.mkdir()
    mapping, rng = {}, np.random.RandomState(0)
    for k, ip in enumerate(ds.original_images_paths):
        aps = []
        for v in range(5):
            size = (32, 32) if (k + v) % 2 else (48, 32)
            ap = folder / f"{k}_{v}.png"
            Image.fromarray(rng.randint(0, 255, size + (3,), np.uint8)).save(ap)
            aps.append(str(ap))
        mapping[os.path.basename(ip)] = aps
    decoded = []
    real = filters._load_u8
    monkeypatch.setattr(filters, "_load_u8", lambda p: (decoded.append(p), real(p))[1])
    monkeypatch.setattr(filters.ops, "h2d", lambda t, dev, dtype=None: t)

    class Sem:
        dev = "cpu"

        def __init__(self):
            self.max_decoded_ahead = 0
            self.seen = 0

        def passes(self, batch):
            self.seen += batch.shape[0]
            self.max_decoded_ahead = max(self.max_decoded_ahead, len(decoded) - self.seen)
            return batch.float().mean((1, 2, 3)).numpy() > 127.0

    sem = Sem()
    out, counters = filters.apply_filters(mapping, ds.original_images_paths, ds, None, semantic=sem, batch_size=3)
    assert sem.seen == 20 and len(decoded) == 20
    assert sem.max_decoded_ahead <= 3, "more than one batch was decoded ahead of the one being evaluated"
    kept = sum(len(v) for v in out.values())
    assert kept + counters["semantic"] == 20 and set(out) == set(mapping)
    for name, aps in out.items():                      # decisions are per image, order preserved
        assert aps == [ap for ap in mapping[name] if np.asarray(Image.open(ap).convert("RGB")).astype(np.float32).mean() > 127.0]


def test_cv_table_cache_is_bounded(monkeypatch):
    monkeypatch.setattr(imageproc.ops, "h2d", lambda t, dev, dtype=None: t)
    monkeypatch.setattr(imageproc, "_CV_DEV_MAX", 4)
    imageproc._CV_DEV.clear()
    for i in range(10):
        imageproc._cv_dev("cpu", ("tap", 100 + i, 64, "lanczos4"), lambda: (np.zeros(4, np.int32), np.zeros(4, np.int16)))
    assert len(imageproc._CV_DEV) == 4
    first = imageproc._cv_dev("cpu", ("tap", 109, 64, "lanczos4"), lambda: (_ for _ in ()).throw(AssertionError("cached entry rebuilt")))
    assert isinstance(first, tuple) and torch.is_tensor(first[0])
    imageproc._CV_DEV.clear()

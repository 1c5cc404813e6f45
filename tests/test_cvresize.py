"""cv2.resize as the reference's `resize_image` calls it (all_utils/utils.py:58-79).
CPU: the host tables of the product (saspa_aug_amd.imageproc) against the oracle's independent restatement, properties of
the oracle (constant images, exact 2x area mean, weights summing to ~2048).  GPU: the kernels bit-exact against the oracle
for every branch (Lanczos4 up-scaling, area down-scaling with fractional and integer factors, the bilinear fallback when
the /64 rounding pushes one side above the source) and through `utils.resize_image` / the run_aug batch path."""
import numpy as np
import pytest
import torch

import saspa_aug_amd  # noqa: F401
from oracle import cv_resize as OR
from saspa_aug_amd import imageproc, utils
from saspa_aug_amd.synthetic import synthetic_image


def test_tables_equal_the_oracle():
    for (s, d) in ((300, 512), (400, 704), (500, 640), (97, 512)):
        o, w = imageproc.cv_tap_tables(s, d, "lanczos4")
        oo, ow = OR.lanczos4_tables(s, d)
        assert np.array_equal(o, oo) and np.array_equal(w, ow)
        assert abs(int(w.sum(1).min()) - 2048) <= 3 and abs(int(w.sum(1).max()) - 2048) <= 3
    for (s, d) in ((700, 704), (512, 576)):
        o, w = imageproc.cv_tap_tables(s, d, "area_linear")
        oo, ow = OR.linear_area_tables(s, d)
        assert np.array_equal(o, oo) and np.array_equal(w, ow) and (w.sum(1) == 2048).all()
    for (s, d) in ((1024, 768), (695, 512), (1000, 512), (513, 512)):
        st, si, al = imageproc.cv_area_tables(s, d)
        tab = OR.area_tab(s, d, s / float(d))
        assert len(tab) == len(si) == st[-1]
        assert [t[1] for t in tab] == si.tolist() and np.array_equal(np.array([t[2] for t in tab], np.float32), al)
        dst = np.repeat(np.arange(d), np.diff(st))
        assert [t[0] for t in tab] == dst.tolist()
        sums = np.add.reduceat(al.astype(np.float64), st[:-1])
        assert np.abs(sums - 1.0).max() < 1e-5


def test_oracle_properties():
    assert np.unique(OR.resize_image(np.full((300, 400, 3), 77, np.uint8), 512)).tolist() == [77]
    assert np.unique(OR.resize_image(np.full((900, 1000, 3), 201, np.uint8), 512)).tolist() == [201]
    img = np.random.RandomState(0).randint(0, 256, (1024, 1024, 3)).astype(np.uint8)
    got = OR.resize_image(img, 512)
    want = (img.reshape(512, 2, 512, 2, 3).astype(int).sum((1, 3)) + 2) >> 2
    assert np.array_equal(got, want)
    img = synthetic_image(300, 400, 1)
    up = OR.resize_image(img, 512)
    assert up.shape == (512, 704, 3) and abs(float(up.mean()) - float(img.mean())) < 1.0


CASES = [(300, 400), (480, 640), (97, 131), (695, 1024), (1200, 1600), (2000, 3000), (1024, 1024), (1536, 2048), (512, 700), (530, 512)]


@pytest.mark.gpu
def test_kernels_bit_exact_vs_oracle(dev):
    for (h, w) in CASES:
        img = np.random.RandomState(h + w).randint(0, 256, (h, w, 3)).astype(np.uint8)
        img[: h // 3] = synthetic_image(h // 3, w, 5)                       # structured content next to noise
        want = OR.resize_image(img, 512)
        got = utils.resize_image(img, 512)
        assert got.shape == want.shape and got.dtype == np.uint8
        assert np.array_equal(got, want), (h, w, int(np.abs(got.astype(int) - want).max()))
    # batched entry: two images in one launch
    a = np.stack([np.random.RandomState(i).randint(0, 256, (300, 400, 3)).astype(np.uint8) for i in range(2)])
    got = imageproc.cv_resize_u8(torch.from_numpy(a).to(dev), 512, 704, "lanczos4").cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], OR.resize_lanczos4(a[i], 512, 704))


@pytest.mark.gpu
def test_run_aug_resizes_sources_on_the_device(dev, tmp_path):
    """Source images that are NOT at the planned size go through the device resize inside the batch; the saved
    `_source.png` is the cv2-semantics result (== oracle), the output has the planned size."""
    from PIL import Image

    from saspa_aug_amd import config as CFG
    from saspa_aug_amd import run_aug as R
    from saspa_aug_amd import weights as W
    from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline
    root = tmp_path / "ds/data"
    prompts = tmp_path / "prompts.txt"
    prompts.write_text("A white airplane on a runway.\nAn airplane above the clouds.\n")
    s = R.Settings(DATASET="synthetic", NUM_PER_IMAGE=1, SEED=1, RESOLUTION=64, BATCH_SIZE=2, PROMPTS_FILE=str(prompts),
                   NUM_INFERENCE_STEPS=2, SEMANTIC_FILTERING=0, MODEL_CONFIDENCE_BASED_FILTERING=0,
                   DATASET_KWARGS=dict(root_path=str(root), n_images=4, sizes=((50, 70), (100, 90), (64, 64), (130, 200)), seed=3))
    cfgs = {k: v for k, v in CFG.tiny().items() if k != "safety"}
    pipe = StableDiffusionControlNetPipeline(W.synth_family(cfgs, seed=3), cfgs).to(dev, torch.float16)
    res = R.main(s, pipe=pipe)
    assert (res["status"] == 1).all()
    out = R.output_folder_for(s, str(root))
    for it in res["items"]:
        raw = np.array(Image.open(it.source_path).convert("RGB"))
        want = OR.resize_image(raw, 64)
        got = np.array(Image.open(f"{out}/{it.image_stem}_source.png"))
        assert got.shape == (it.height, it.width, 3) and np.array_equal(got, want), it.source_path
        assert np.array(Image.open(it.output_path)).shape == (it.height, it.width, 3)

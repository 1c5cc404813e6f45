"""Filter stage, CPU side: the oracle restatement of WSDAN_CAL against logits produced by the REFERENCE's own
module (tests/golden/make_filter_golden.py), architecture pins of CLIP RN50 (public parameter count), BatchNorm folding, the
OpenAI -> transformers key conversion of the text tower, the PIL-exact resample tables, class-id tables, and the host logic
of `apply_filters` / `create_json_of_image_name_to_augmented_images_paths` with stand-in filter models."""
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

import saspa_aug_amd  # noqa: F401
from oracle import filter_models as FM
from oracle import sd_models as OM
from saspa_aug_amd import config as CFG
from saspa_aug_amd import dataset_utils as DU
from saspa_aug_amd import filters, imageproc, utils
from saspa_aug_amd import weights as W

G = json.load(open(Path(__file__).parent / "golden" / "reference_filter_golden.json"))


@pytest.mark.parametrize("tag", ["resnet50", "resnet101"])
def test_oracle_cal_equals_the_reference_module(tag):
    g = G[tag]
    cfg = dict(g["cfg"], layers=tuple(g["cfg"]["layers"]))
    sd = W.synth_state_dict("cal", cfg, g["weight_seed"])
    x = torch.randn(tuple(g["input_shape"]), generator=torch.Generator().manual_seed(g["input_seed"]))
    with torch.no_grad():
        got = FM.wsdan_cal_logits(sd, cfg, x)
    ref = torch.tensor(g["logits"])
    assert (got.double() - ref).abs().max().item() < 2e-4 * ref.abs().max().item(), (got, ref)
    assert torch.equal(got.argsort(dim=-1, descending=True)[:, :5], ref.argsort(dim=-1, descending=True)[:, :5])


def test_clip_rn50_parameter_count_is_the_public_one():
    n = sum((math.prod(sh) if len(sh) else 1) for name, sh, _ in W.clip_rn50_spec(CFG.CLIP_RN50) if "running" not in name)
    assert n == 102_007_137                                   # OpenAI CLIP RN50 (102 M parameters)


def test_bn_fold_and_text_key_conversion():
    cf = CFG.tiny_filters()["clip_rn50"]
    sd = W.synth_state_dict("clip_rn50", cf, 3)
    x = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(1))
    w, b = W.fold_bn(sd["visual.conv1.weight"], sd, "visual.bn1")
    ref = FM.bn(sd, "visual.bn1", torch.nn.functional.conv2d(x, sd["visual.conv1.weight"], stride=2, padding=1))
    got = torch.nn.functional.conv2d(x, w, b, stride=2, padding=1)
    assert (got - ref).abs().max() < 1e-5
    # OpenAI text tower == the transformers-named tower after conversion (final-LN state at the EOT token, projected)
    ids = torch.randint(1, cf["vocab"] - 2, (3, 77), generator=torch.Generator().manual_seed(2))
    ids[:, 0] = cf["vocab"] - 2
    for r, e in enumerate((5, 9, 30)):
        ids[r, e] = cf["vocab"] - 1
        ids[r, e + 1:] = 0
    ref = FM.clip_openai_text(sd, cf, ids)
    hf = W.openai_clip_text_to_hf(sd, cf["text_layers"])
    tcfg = dict(vocab=cf["vocab"], width=cf["text_width"], layers=cf["text_layers"], heads=cf["text_heads"], mlp=4 * cf["text_width"], max_pos=77)
    last = OM.clip_text_forward(hf, tcfg, ids)
    got = torch.nn.functional.linear(last[torch.arange(3), ids.argmax(-1)], hf["text_projection.weight"])
    assert (got - ref).abs().max() < 1e-4 * ref.abs().max()


@pytest.mark.parametrize("filt,pil", [("bilinear", Image.BILINEAR), ("bicubic", Image.BICUBIC)])
def test_resample_tables_reproduce_pillow(filt, pil):
    """The coefficient tables the device kernel consumes, applied on the host in Pillow's integer arithmetic, equal
    PIL.Image.resize bit for bit (down- and up-scaling, both filters)."""
    rng = np.random.RandomState(0)
    for (h, w, oh, ow) in ((64, 96, 28, 40), (37, 53, 64, 80), (512, 512, 256, 256)):
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        want = np.asarray(Image.fromarray(img).resize((ow, oh), pil))

        def one_pass(a, axis, out_len):
            in_len = a.shape[axis]
            bounds, coeffs, _ = imageproc.resample_tables(in_len, out_len, 0, None, filt)
            a = np.moveaxis(a, axis, 0).astype(np.int64)
            out = np.zeros((out_len,) + a.shape[1:], np.int64)
            for r in range(out_len):
                x0, n = bounds[r]
                acc = (coeffs[r, :n, None, None].astype(np.int64) * a[x0:x0 + n]).sum(0) + (1 << (imageproc.PRECISION_BITS - 1))
                out[r] = np.clip(acc >> imageproc.PRECISION_BITS, 0, 255)
            return np.moveaxis(out, 0, axis).astype(np.uint8)
        got = one_pass(one_pass(img, 1, ow), 0, oh)
        assert np.array_equal(got, want), (filt, h, w, oh, ow, np.abs(got.astype(int) - want).max())


def test_class_id_tables(tmp_path, monkeypatch):
    import sys
    sys.path.insert(0, str(Path(__file__).parent / "golden"))
    import dataset_fixtures as FX
    monkeypatch.chdir(tmp_path)
    quiet = lambda *a, **k: None   # noqa: E731
    FX.build_cub(tmp_path), FX.build_cars(tmp_path), FX.build_dtd(tmp_path)
    cub = DU.CUBUtils(print_func=quiet)
    t = cub.get_image_path_to_class_id_dict()
    assert all(t[p] == int(Path(p).parent.name.split(".")[0]) - 1 for p in cub.original_images_paths)
    cars = DU.CarsUtils(print_func=quiet)
    t = cars.get_image_path_to_class_id_dict()
    assert set(cars.original_images_paths) <= set(t) and set(t.values()) == {0, 1, 2}
    dtd = DU.DTDUtils(print_func=quiet)
    t = dtd.get_image_path_to_class_id_dict()
    assert {t[p] for p in dtd.original_images_paths} == {0, 1, 2} and t[dtd.original_images_paths[0]] == 0
    syn = DU.SyntheticUtils(root_path=str(tmp_path / "syn/data"), n_images=6, sizes=((32, 32),), print_func=quiet)
    t = syn.get_image_path_to_class_id_dict()
    assert set(syn.original_images_paths) == set(t) and max(t.values()) == len(set(DU.SyntheticUtils.VARIANTS)) - 1


class _Fake:
    """Stand-in filter: decides from the mean grey level of the image."""

    def __init__(self, fn):
        self.fn = fn

    def passes(self, batch, labels=None):
        m = batch.float().mean(dim=(1, 2, 3)).numpy()
        return np.array([self.fn(v, None if labels is None else labels[i]) for i, v in enumerate(m)])


def test_apply_filters_and_json_host_logic(tmp_path, monkeypatch):
    """Confidence filter first, semantic second, per augmented image; every original keeps its key; the JSON carries the
    filtered lists under the filtered file name (all_utils/utils.py:357-366, :401-409, :437-443)."""
    root = tmp_path / "ds/data"
    ds = DU.SyntheticUtils(root_path=str(root), n_images=3, sizes=((16, 16),), print_func=lambda *a, **k: None)
    folder = root / "aug_data/controlnet/sd_v1.5/canny/run_seed_1/images"
    folder.mkdir(parents=True)
    stems = [Path(p).stem for p in ds.original_images_paths]
    levels = {0: [10, 200], 1: [120, 130], 2: [250]}
    for k, stem in enumerate(stems):
        for v, lv in enumerate(levels[k]):
            Image.fromarray(np.full((16, 16, 3), lv, np.uint8)).save(folder / f"{stem}_prompt_An airplane_{v}.png")
        Image.fromarray(np.full((16, 16, 3), 7, np.uint8)).save(folder / f"{stem}_source.png")
    monkeypatch.setattr(filters.ops, "h2d", lambda t, dev, dtype=None: t)
    conf = _Fake(lambda v, lb: v > 50)             # drops the level-10 image
    sem = _Fake(lambda v, lb: v < 240)             # drops the level-250 image
    jp = utils.create_json_of_image_name_to_augmented_images_paths(
        ds, str(folder), semantic_filtering=1, model_confidence_based_filtering=1, init_log=False,
        original_images_paths=ds.original_images_paths, min_files=1, filter_models=(sem, conf), device="cpu")
    assert Path(jp).name == "semantic_filtering-model_confidence_based_filtering_top_10_classes-aug.json"
    body = json.load(open(jp))
    assert list(body) == [Path(p).name for p in ds.original_images_paths]
    got = {k: sorted(Path(p).name for p in v) for k, v in body.items()}
    assert got == {f"{stems[0]}.png": [f"{stems[0]}_prompt_An airplane_1.png"],
                   f"{stems[1]}.png": [f"{stems[1]}_prompt_An airplane_0.png", f"{stems[1]}_prompt_An airplane_1.png"],
                   f"{stems[2]}.png": []}
    mapping = utils.match_augmented_images(ds.original_images_paths, sorted(p.name for p in folder.iterdir()), str(folder))
    _, counters = filters.apply_filters(mapping, ds.original_images_paths, ds, "cpu", sem, conf)
    assert counters == dict(not_in_top_k=1, semantic=1)
    # without a GPU and without injected models the filtered JSON is refused, never written unfiltered
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            utils.create_json_of_image_name_to_augmented_images_paths(
                ds, str(folder), semantic_filtering=1, model_confidence_based_filtering=1, init_log=False,
                original_images_paths=ds.original_images_paths, min_files=1)

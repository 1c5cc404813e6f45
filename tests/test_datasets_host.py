"""Per-dataset plumbing of the generation loop (CUB / Stanford Cars / DTD / CompCars car-parts) against goldens written by the
REFERENCE's own classes on the same fixture trees (tests/golden/make_dataset_golden.py, build container only), and the
loop's per-dataset prompt branches (run_aug/run_aug.py:361-363, :386-427) against a hand replay with the reference's prompt
files -- which ship here as data and must be byte-identical."""
import json
import os
import random
import sys
from pathlib import Path

import numpy as np
import pytest

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import dataset_utils as DU
from saspa_aug_amd import run_aug as R

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE / "golden"))
import dataset_fixtures as FX  # noqa: E402

G = json.load(open(HERE / "golden" / "reference_dataset_golden.json"))
quiet = lambda *a, **k: None   # noqa: E731


@pytest.fixture()
def tree(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)          # the reference's default roots are relative to the CWD
    return tmp_path


def _check(ds, g, keyed_by_stem):
    assert ds.name == g["name"] and ds.meta_class == g["meta_class"] and str(ds.root_path) == g["root_path"]
    assert sorted(ds.original_images_paths) == g["original_images_paths"]
    assert dict(sorted(ds.get_image_stem_to_class_str_dict().items())) == g["class_dict"]
    first = sorted(ds.original_images_paths)[0]
    assert sorted(ds.get_image_path_with_same_class(first)) == g["same_class_of_first"]
    assert ds.get_basic_prompt() == g["basic_prompt"]
    assert sorted(map(str, ds.get_classes())) == g["classes"]


def test_cub(tree):
    FX.build_cub(tree)
    ds = DU.CUBUtils(print_func=quiet)
    _check(ds, G["cub"], False)
    assert ds.original_images_paths == G["cub_order"]                       # file order feeds the RNG replay
    assert sorted(DU.CUBUtils(split="val", print_func=quiet).original_images_paths) == G["cub_val"]


def test_cars(tree):
    FX.build_cars(tree)
    _check(DU.CarsUtils(print_func=quiet), G["cars"], True)
    assert sorted(DU.CarsUtils(split="val", print_func=quiet).original_images_paths) == G["cars_val"]


def test_dtd(tree):
    FX.build_dtd(tree)
    ds = DU.DTDUtils(print_func=quiet)
    _check(ds, G["dtd"], False)
    assert ds.original_images_paths == G["dtd_order"]


def test_compcars_parts(tree):
    FX.build_compcars(tree)
    ds = DU.CompCarsPartsUtils(print_func=quiet)
    g = G["compcars-parts"]
    assert (ds.name, ds.meta_class, len(ds.original_images_paths)) == (g["name"], g["meta_class"], g["n_original"])
    assert ds.original_images_paths[:25] == g["first_paths"]
    d = ds.get_image_stem_to_class_str_dict()
    assert {p: d[p] for p in g["first_paths"]} == g["class_of_first"]
    assert sorted(ds.get_image_path_with_same_class(g["first_paths"][0])) == g["same_class_of_first"]
    assert ds.get_basic_prompt() == g["basic_prompt"] and {k: ds.get_basic_prompt(part=k) for k in "1234"} == g["part_prompts"]
    assert len(ds.get_classes()) == g["n_classes"]
    assert len(DU.CompCarsPartsUtils(split="val", print_func=quiet).original_images_paths) == g["n_val"]


def test_missing_dataset_fails_loudly(tree):
    for cls in (DU.CUBUtils, DU.CarsUtils, DU.DTDUtils, DU.CompCarsPartsUtils):
        with pytest.raises(FileNotFoundError):
            cls(print_func=quiet)


def _replay(s, paths, class_dict, prompts=None, captions=None, ds_utils=None, out="out"):
    R.utils.set_seed(1)
    items = R.plan_work(s, paths, prompts, out, class_dict, image_size_fn=lambda p: (512, 512), ds_utils=ds_utils, captions=captions)
    return dict(prompts=[it.prompt for it in items], file_names=[Path(it.output_path).name for it in items],
                py_random_after=random.random(), np_random_after=float(np.random.rand()))


def _settings(dataset, **kw):
    base = dict(DATASET=dataset, NUM_PER_IMAGE=2, SEED=1, USE_ARTISTIC_PROMPTS=False, PROMPT_WITH_SUB_CLASS=True)
    base.update(kw)
    return R.Settings(**base)


def test_prompt_branches_replay_the_reference(tree):
    s = _settings("cub")
    assert _replay(s, G["cub_order"], G["cub"]["class_dict"], prompts=R.read_prompts(R.default_prompts_file(s))) == G["replay_cub"]
    s = _settings("cars", USE_ARTISTIC_PROMPTS=True)
    assert _replay(s, G["cars"]["original_images_paths"], G["cars"]["class_dict"],
                   prompts=R.read_prompts(R.default_prompts_file(s))) == G["replay_cars"]
    s = _settings("dtd", PROMPT_TYPE="captions")
    caps = json.load(open(R.default_prompts_file(s)))
    assert _replay(s, G["dtd_order"], G["dtd"]["class_dict"], captions=caps) == G["replay_dtd"]
    s = _settings("compcars-parts")
    cpp = G["compcars-parts"]

    class Parts:
        @staticmethod
        def get_basic_prompt(part=None):
            return cpp["part_prompts"][str(part)]
    assert _replay(s, cpp["first_paths"][:6], cpp["class_of_first"], prompts=R.read_prompts(R.default_prompts_file(s)),
                   ds_utils=Parts) == G["replay_compcars"]


def test_prompt_file_selection():
    sel = {d: Path(R.default_prompts_file(_settings(d))).name for d in ("planes", "synthetic", "cars", "compcars-parts", "cub")}
    assert sel == {"planes": "planes-100-gpt_v1.txt", "synthetic": "planes-100-gpt_v1.txt", "cars": "cars-100-gpt_v1.txt",
                   "compcars-parts": "cars-100-gpt_v1.txt", "cub": "cub-100-gpt_v1.txt"}            # run_aug/run_aug.py:589-666
    assert Path(R.default_prompts_file(_settings("dtd", PROMPT_TYPE="captions"))).name == "dtd_captions.json"
    with pytest.raises(NotImplementedError):
        R.default_prompts_file(_settings("dtd"))
    for d, n in (("planes", 100), ("cars", 100), ("cub", 100)):
        assert len(R.read_prompts(R.default_prompts_file(_settings(d)))) == n


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="build container only")
def test_shipped_data_files_equal_the_reference():
    pkg = Path(DU.__file__).parent
    pairs = [("prompts_engineering/gpt_prompts/planes-100-gpt_v1.txt",) * 2, ("prompts_engineering/gpt_prompts/cars-100-gpt_v1.txt",) * 2,
             ("prompts_engineering/gpt_prompts/cub-100-gpt_v1.txt",) * 2, ("prompts_engineering/captions/dtd_captions.json",) * 2,
             ("datasets_files/cub_val.txt", "fgvc/datasets_files/cub_val.txt"), ("datasets_files/cars_val.txt", "fgvc/datasets_files/cars_val.txt"),
             ("datasets_files/compcars-parts/train.csv", "fgvc/datasets_files/compcars-parts/train.csv")]
    for ours, ref in pairs:
        assert (pkg / ours).read_bytes() == (Path("/root/reference") / ref).read_bytes(), ours

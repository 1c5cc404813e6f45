#!/usr/bin/env python3
"""Generates tests/golden/reference_host_golden.json by IMPORTING the reference's pure-Python
host functions in the build container (the reference never travels to the GPU box; only this
script and the JSON it writes are committed).

    python tests/golden/make_golden.py            # needs /root/reference

What is captured (SURVEY 8c "what can be imported here"):
  * all_utils/utils.py  get_aug_json_path / HWC3 / resize_image's target-size arithmetic
    (cv2.resize stubbed to record dsize + interpolation) / create_json_of_image_name_to_
    augmented_images_paths with every filter off, run on a fixture folder
  * fgvc/datasets/aug_wrapper_dataset.py  AugWrapperDataset.init_augmentation / get_aug_image
    consuming that JSON (the downstream contract)
  * prompts_engineering constants, and the first file names of a seed-1 planes run obtained by
    replaying run_aug/run_aug.py:380-429 by hand (same RNG call sequence) on 100 stand-in prompts.
Third-party modules the reference imports but this path never calls (cv2, clip, lpips,
torchvision, wandb, matplotlib ...) are stubbed in sys.modules."""
import importlib
import importlib.util
import json
import os
import random
import sys
import tempfile
import types
from pathlib import Path
from unittest import mock

import numpy as np

REF = "/root/reference"
OUT = Path(__file__).parent / "reference_host_golden.json"


class _ClassyModule(types.ModuleType):
    """Stub module whose every attribute is a plain class (usable as a base class)."""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def stub_modules():
    for name in ["torchvision.datasets", "torchvision.datasets.folder", "torchvision.datasets.utils"]:
        sys.modules[name] = _ClassyModule(name)
    for name in ["cv2", "clip", "clip.clip", "lpips", "torchvision", "torchvision.transforms", "torchvision.datasets",
                 "torchvision.datasets.utils", "torchvision.datasets.folder", "torchvision.models", "wandb", "matplotlib",
                 "matplotlib.pyplot", "cutmix", "cutmix.cutmix", "cutmix.utils", "scipy.io", "timm", "seaborn", "sklearn",
                 "sklearn.metrics", "sklearn.manifold"]:
        if name not in sys.modules:
            m = mock.MagicMock(name=name)
            m.__path__ = []
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = m


def main():
    sys.path.insert(0, REF)
    stub_modules()
    golden = {}

    import cv2  # the stub
    calls = []

    def fake_resize(img, dsize, interpolation=None):
        calls.append((tuple(dsize), "LANCZOS4" if interpolation is cv2.INTER_LANCZOS4 else "AREA"))
        return np.zeros((dsize[1], dsize[0], 3), np.uint8)
    cv2.resize = fake_resize
    from all_utils import utils as RU

    # ---- get_aug_json_path ----
    cases = [dict(), dict(semantic_filtering=1, model_confidence_based_filtering=1), dict(semantic_filtering=1),
             dict(model_confidence_based_filtering=1, conf_top_k=5), dict(model_confidence_based_filtering=1, filter_confidence_higher_than=3)]
    folder = "data/FGVC-Aircraft/fgvc-aircraft-2013b/data/aug_data/controlnet/sd_v1.5/canny/gpt-meta_class_prompt_w_sub_class_artistic_prompts_p_0.5_seed_1/images"
    golden["aug_json_path"] = [dict(kwargs=c, path=RU.get_aug_json_path(folder, **c)) for c in cases]

    # ---- resize target sizes ----
    sizes = [(512, 512), (695, 1024), (1024, 695), (300, 400), (2000, 3000), (480, 640), (1200, 1600), (333, 1000), (64, 64), (768, 512)]
    rs = []
    for h, w in sizes:
        calls.clear()
        RU.resize_image(np.zeros((h, w, 3), np.uint8), 512)
        (tw, th), interp = calls[0]
        rs.append(dict(h=h, w=w, res=512, out_h=th, out_w=tw, interp=interp))
    golden["resize_targets"] = rs

    # ---- HWC3 ----
    rng = np.random.RandomState(0)
    g1 = rng.randint(0, 256, (3, 4)).astype(np.uint8)
    g4 = rng.randint(0, 256, (3, 4, 4)).astype(np.uint8)
    golden["hwc3"] = dict(gray_in=g1.tolist(), gray_out=RU.HWC3(g1).tolist(), rgba_in=g4.tolist(), rgba_out=RU.HWC3(g4).tolist())

    # ---- prompt constants ----
    import prompts_engineering as PE
    golden["ARTISTIC_PROMPTS"] = list(PE.ARTISTIC_PROMPTS)
    golden["IMAGE_VARIATIONS_PROMPTS"] = list(PE.IMAGE_VARIATIONS_PROMPTS)

    # ---- hand replay of the seed-1 planes loop (run_aug.py:305-309, :380-429) with the reference's
    # ARTISTIC_PROMPTS; the 100 prompts are synthetic stand-ins of the same count (the RNG call
    # sequence, not the prompt text, is what is being pinned) ----
    prompts = [f"A white airplane number {k} flying over terrain type {k % 7}."[:150] for k in range(100)]
    golden["replay_prompts"] = prompts
    random.seed(1)
    np.random.seed(1)
    stems = [f"{1000000 + i:07d}" for i in range(6)]
    classes = {s: ["Boeing 707-320", "Airbus A320", "Cessna 172"][i % 3] for i, s in enumerate(stems)}
    names = []
    for stem in stems:
        ps = [p[:-1] if p[-1] == "." else p for p in prompts]
        sampled = np.random.choice(ps, 2)
        for i, prompt in enumerate(sampled):
            if True and ((i % 2 == 0 and 0.5 == 0.5) or (random.random() < 0.5 and 0.5 != 0.5)):
                prompt = f"{prompt}, {np.random.choice(PE.ARTISTIC_PROMPTS)}"
            prompt = prompt.replace("airplane", f"{classes[stem]} airplane")
            names.append(f"{stem[:40]}_prompt_{prompt.replace('/', '-')}_{i}.png")
    golden["planes_seed1_replay"] = dict(stems=stems, classes=classes, file_names=names, py_random_after=random.random(),
                                         np_random_after=float(np.random.rand()))

    # ---- JSON body from the reference's own create_json on a fixture folder (filters off) ----
    with tempfile.TemporaryDirectory() as td:
        cwd = os.getcwd()
        os.chdir(td)
        try:
            root = Path("data/FGVC-Aircraft/fgvc-aircraft-2013b/data")
            (root / "images").mkdir(parents=True)
            ids = [f"{1000000 + i:07d}" for i in range(5)]
            (root / "images_train.txt").write_text("\n".join(ids) + "\n")
            (root / "images_manufacturer_train.txt").write_text("".join(f"{i} Boeing\n" for i in ids))
            (root / "images_variant_train.txt").write_text("".join(f"{i} 707-320\n" for i in ids))
            imgs = root / "aug_data/controlnet/sd_v1.5/canny/run_seed_1/images"
            imgs.mkdir(parents=True)
            from PIL import Image
            listing = []
            for k, i in enumerate(ids[:4]):                  # the 5th original has no augmentations
                for v in range(2 if k != 2 else 3):
                    listing.append(f"{i}_prompt_An airplane, a painting of monet_{v}.png")
                listing += [f"{i}_source.png", f"{i}_control.png"]
            for n in listing:
                Image.fromarray(np.full((8, 8, 3), 90, np.uint8)).save(imgs / n)
            import all_utils.dataset_utils as RDU
            RDU.PlanesUtils.download_torchvision_dataset_if_needed = lambda self, *a, **k: None
            jp = RU.create_json_of_image_name_to_augmented_images_paths(
                "planes", str(imgs), semantic_filtering=False, model_confidence_based_filtering=False, init_log=False)
            body = json.load(open(jp))
            golden["create_json"] = dict(ids=ids, listing=listing, images_dir=str(imgs), json_path=jp,
                                         body={k: sorted(v) for k, v in body.items()})

            # ---- the downstream consumer on that JSON ----
            spec = importlib.util.spec_from_file_location("awd", os.path.join(REF, "fgvc/datasets/aug_wrapper_dataset.py"))
            awd = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(awd)
            ds = object.__new__(awd.AugWrapperDataset)
            ds.print_func = lambda *a, **k: None
            ds._image_files = [str(root / "images" / f"{i}.jpg") for i in ids]
            ds._labels = list(range(len(ids)))
            ds.original_data_length = len(ids)
            sorted_json = Path(td) / "sorted.json"
            json.dump({k: sorted(v) for k, v in body.items()}, open(sorted_json, "w"))
            ds.init_augmentation(str(sorted_json), 0.5, 2)
            random.seed(7)
            picks = [ds.get_aug_image(p, idx) for idx, p in enumerate(ds._image_files * 3)]
            golden["consumer"] = dict(aug_sample_ratio=0.5, limit_aug_per_image=2, seed=7, picks=picks,
                                      kept_keys=sorted(ds.aug_json.keys()))
        finally:
            os.chdir(cwd)

    json.dump(golden, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generates tests/golden/reference_dataset_golden.json by running the REFERENCE's dataset classes
(all_utils/dataset_utils.py: CUBUtils :448, CarsUtils :227, DTDUtils :302, CompCarsPartsUtils :342) on the fixture trees of
dataset_fixtures.py, plus a hand replay of the loop's per-dataset prompt branches (run_aug/run_aug.py:361-363, :386-427)
with the reference's own prompt files.  Build container only (/root/reference); the JSON is what is committed.

    python tests/golden/make_dataset_golden.py"""
import json
import os
import random
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
REF = "/root/reference"
OUT = HERE / "reference_dataset_golden.json"


def main():
    import make_golden as MG
    import scipy.io  # noqa: F401   (the real module: the dataset classes parse .mat files)
    sys.path.insert(0, REF)
    MG.stub_modules()
    import dataset_fixtures as FX
    golden = {}
    with tempfile.TemporaryDirectory() as td:
        cwd = os.getcwd()
        os.chdir(td)
        try:
            FX.build_cub(td), FX.build_cars(td), FX.build_dtd(td), FX.build_compcars(td)
            import all_utils.dataset_utils as RDU
            for cls in (RDU.PlanesUtils, RDU.DTDUtils):
                cls.download_torchvision_dataset_if_needed = lambda self, *a, **k: None
            quiet = lambda *a, **k: None   # noqa: E731

            def dump(ds, keyed_by_stem):
                d = ds.get_image_stem_to_class_str_dict() if keyed_by_stem else ds.get_image_path_to_class_str_dict()
                src = ds.original_images_paths[0]
                return dict(name=ds.name, meta_class=ds.meta_class, root_path=str(ds.root_path),
                            original_images_paths=sorted(ds.original_images_paths), class_dict=dict(sorted(d.items())),
                            same_class_of_first=sorted(ds.get_image_path_with_same_class(sorted(ds.original_images_paths)[0])),
                            basic_prompt=ds.get_basic_prompt(), classes=sorted(map(str, ds.get_classes())))
            golden["cub"] = dump(RDU.CUBUtils(print_func=quiet), False)
            golden["cub_val"] = sorted(RDU.CUBUtils(split="val", print_func=quiet).original_images_paths)
            golden["cub_order"] = RDU.CUBUtils(print_func=quiet).original_images_paths            # file order matters (RNG replay)
            golden["cars"] = dump(RDU.CarsUtils(print_func=quiet), True)
            golden["cars_val"] = sorted(RDU.CarsUtils(split="val", print_func=quiet).original_images_paths)
            golden["dtd"] = dump(RDU.DTDUtils(print_func=quiet), False)
            golden["dtd_order"] = RDU.DTDUtils(print_func=quiet).original_images_paths
            cc = RDU.CompCarsPartsUtils(print_func=quiet)
            d = cc.get_image_path_to_class_str_dict()
            some = cc.original_images_paths[:25]
            golden["compcars-parts"] = dict(
                name=cc.name, meta_class=cc.meta_class, n_original=len(cc.original_images_paths), first_paths=some,
                class_of_first={p: d[p] for p in some}, same_class_of_first=sorted(cc.get_image_path_with_same_class(some[0])),
                basic_prompt=cc.get_basic_prompt(), part_prompts={k: cc.get_basic_prompt(part=k) for k in "1234"},
                n_classes=len(cc.get_classes()), n_val=len(RDU.CompCarsPartsUtils(split="val", print_func=quiet).original_images_paths))
        finally:
            os.chdir(cwd)

    # ---- per-dataset prompt branches: hand replay of run_aug/run_aug.py:361-363, :380-429 ----
    import prompts_engineering as PE

    def replay(dataset, paths, class_dict, prompts=None, captions=None, artistic=False, part_prompt=None, num=2):
        random.seed(1)
        np.random.seed(1)
        names, out_prompts = [], []
        for src in paths:
            stem = Path(src).stem
            if captions is not None:
                ps = [captions[src]["caption"]] * num
                ps = [p[:150] for p in ps]
            else:
                ps = prompts
            ps = [p[:-1] if p[-1] == "." else p for p in ps]
            sampled = np.random.choice(ps, num)
            for i, prompt in enumerate(sampled):
                if dataset == "compcars-parts":
                    prompt = f"{part_prompt(src.split('/')[-2])} {prompt}"
                if artistic and ((i % 2 == 0 and 0.5 == 0.5) or (random.random() < 0.5 and 0.5 != 0.5)):
                    prompt = f"{prompt}, {np.random.choice(PE.ARTISTIC_PROMPTS)}"
                if dataset == "cars":
                    prompt = prompt.replace("car", f"{class_dict[stem]} car")
                elif dataset == "dtd":
                    prompt = f"{prompt} with a {class_dict[src]} texture"
                elif dataset == "compcars-parts":
                    prompt = prompt.replace("car", f"{class_dict[src]} car")
                elif dataset == "cub":
                    prompt = prompt.replace("bird", f"{class_dict[src]} bird")
                out_prompts.append(str(prompt))
                names.append(f"{stem[:40]}_prompt_{prompt.replace('/', '-')}_{i}.png")
        return dict(prompts=out_prompts, file_names=names, py_random_after=random.random(), np_random_after=float(np.random.rand()))

    def read(f):
        return [p.strip()[:150] for p in open(os.path.join(REF, "prompts_engineering/gpt_prompts", f)).readlines()]
    g = golden
    g["replay_cub"] = replay("cub", g["cub_order"], g["cub"]["class_dict"], prompts=read("cub-100-gpt_v1.txt"))
    g["replay_cars"] = replay("cars", g["cars"]["original_images_paths"], g["cars"]["class_dict"], prompts=read("cars-100-gpt_v1.txt"),
                              artistic=True)
    caps = json.load(open(os.path.join(REF, "prompts_engineering/captions/dtd_captions.json")))
    g["replay_dtd"] = replay("dtd", g["dtd_order"], g["dtd"]["class_dict"], captions=caps)
    cpp = g["compcars-parts"]
    g["replay_compcars"] = replay("compcars-parts", cpp["first_paths"][:6], cpp["class_of_first"], prompts=read("cars-100-gpt_v1.txt"),
                                  part_prompt=lambda k: cpp["part_prompts"][k])
    json.dump(golden, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

"""Tiny on-disk stand-ins with the file layout of the datasets the generation loop runs on (CUB-200-2011, Stanford Cars,
DTD, CompCars car-parts): empty image files, real annotation formats.  Shared by tests/golden/make_dataset_golden.py (which
runs the REFERENCE's dataset classes on them, in the build container) and tests/test_datasets_host.py (which runs ours on
the same trees): fixtures are data, not reference code."""
import os
from pathlib import Path

import numpy as np

PKG_FILES = Path(__file__).resolve().parent.parent.parent / "saspa-aug_amd" / "datasets_files"


def _touch(p):
    p.parent.mkdir(parents=True, exist_ok=True)
    p.write_bytes(b"")


def build_cub(cwd):
    root = Path(cwd) / "data/CUB/CUB_200_2011"
    val = [ln.strip() for ln in open(PKG_FILES / "cub_val.txt")][:3]            # real validation entries -> must be filtered
    classes = ["001.Black_footed_Albatross", "002.Laysan_Albatross"] + sorted({v.split("/")[0] for v in val})
    rels = []
    for c in classes[:2]:
        rels += [f"{c}/{c.split('.')[1]}_{k:04d}_{100 + k}.jpg" for k in range(1, 5)]
    rels += val
    (root / "images").mkdir(parents=True, exist_ok=True)
    all_classes = [f"{i + 1:03d}.Class_{i + 1}" for i in range(200)]
    for c in classes:
        all_classes[int(c.split(".")[0]) - 1] = c
    (root / "classes.txt").write_text("".join(f"{i + 1} {c}\n" for i, c in enumerate(all_classes)))
    with open(root / "images.txt", "w") as fi, open(root / "image_class_labels.txt", "w") as fl, \
            open(root / "train_test_split.txt", "w") as fs:
        for k, rel in enumerate(rels):
            _touch(root / "images" / rel)
            fi.write(f"{k + 1} {rel}\n")
            fl.write(f"{k + 1} {int(rel.split('.')[0])}\n")
            fs.write(f"{k + 1} {0 if k % 4 == 3 else 1}\n")                      # every 4th image is a test image
    return root


def build_cars(cwd):
    import scipy.io as sio
    root = Path(cwd) / "data/stanford_cars/stanford_cars"
    val = [ln.strip() for ln in open(PKG_FILES / "cars_val.txt")][:2]
    names = ["AM General Hummer SUV 2000", "Acura RL Sedan 2012", "Audi S4 Sedan 2012"]
    files = [f"{k:05d}.jpg" for k in (1, 2, 3, 4, 5, 6)] + val
    (root / "devkit").mkdir(parents=True, exist_ok=True)
    cn = np.empty((1, len(names)), dtype=object)
    for i, n in enumerate(names):
        cn[0, i] = np.array([n])
    sio.savemat(root / "devkit/cars_meta.mat", {"class_names": cn})
    dt = [(k, "O") for k in ("bbox_x1", "bbox_y1", "bbox_x2", "bbox_y2", "class", "fname")]
    ann = np.empty((1, len(files)), dtype=dt)
    for i, f in enumerate(files):
        _touch(root / "cars_train" / f)
        ann[0, i] = (np.array([[1]]), np.array([[2]]), np.array([[30]]), np.array([[40]]), np.array([[i % 3 + 1]]), np.array([f]))
    sio.savemat(root / "devkit/cars_train_annos.mat", {"annotations": ann})
    return root


def build_dtd(cwd):
    root = Path(cwd) / "data/DTD/dtdataset/dtd"
    # file names that own a caption in the shipped dtd_captions.json (the loop indexes the captions by image path)
    import json
    caps = json.load(open(PKG_FILES.parent / "prompts_engineering" / "captions" / "dtd_captions.json"))
    rels = []
    for t in ("banded", "blotchy", "woven"):
        rels += sorted(k.split("/images/")[1] for k in caps if f"/images/{t}/" in k)[:3]
    for rel in rels:
        _touch(root / "images" / rel)
    (root / "labels").mkdir(parents=True, exist_ok=True)
    (root / "labels/train1.txt").write_text("\n".join(r for i, r in enumerate(rels) if i % 3 != 2) + "\n")
    return root


def build_compcars(cwd, n_rows=40):
    import scipy.io as sio
    root = Path(cwd) / "data/compcars"
    rows = [ln.strip().split(",")[0] for ln in open(PKG_FILES / "compcars-parts" / "train_and_test.csv")]
    for rel in rows:                                                              # every make/model folder the lists name
        os.makedirs(root / "part" / Path(rel).parent.parent.parent, exist_ok=True)
    for rel in rows[:n_rows]:
        _touch(root / "part" / rel)
    n_make = max(int(r.split("/")[0]) for r in rows)
    n_model = max(int(r.split("/")[1]) for r in rows)
    mk = np.empty((n_make, 1), dtype=object)
    md = np.empty((n_model, 1), dtype=object)
    for i in range(n_make):
        mk[i, 0] = np.array([f"Make{i + 1}"]) if i % 17 != 5 else np.array([])   # the real table has empty cells
    for i in range(n_model):
        md[i, 0] = np.array([f"Model {i + 1}"]) if i % 29 != 7 else np.array([])
    (root / "misc").mkdir(parents=True, exist_ok=True)
    sio.savemat(root / "misc/make_model_name.mat", {"make_names": mk, "model_names": md})
    return root

"""Per-dataset path / class bookkeeping touched by the generation loop (mirror of the thin
part of the reference's all_utils/dataset_utils.py: `original_images_paths`, `root_path`,
`meta_class`, `get_image_stem_to_class_str_dict`, `get_image_path_with_same_class`,
`get_basic_prompt`).  The torchvision downloads, the other dataset
families' CSV plumbing and the baseline-classifier loader are out of scope.

`SyntheticUtils` builds an on-disk dataset with the FGVC-Aircraft file layout from seeded
synthetic images, for the boxes that have no datasets (tests / bench / demo runs)."""
import os
from pathlib import Path

from . import utils

DATASETS_SUPPORTED = ["planes", "synthetic"]


class BaseUtils:
    def __init__(self, split="train", root_path: str = "", print_func=print):
        self.name = ""
        self.meta_class = ""
        self.root_path = Path(root_path)
        self.split = split
        self.print_func = print_func
        self.original_images_paths = []
        self.image_path_to_class_str_dict = {}

    def get_classes(self):
        return sorted(set(self.image_path_to_class_str_dict.values()))

    @property
    def num_classes(self):
        return len(self.get_classes())

    def get_image_stem_to_class_str_dict(self):
        raise NotImplementedError

    def get_basic_prompt(self):
        raise NotImplementedError

    def get_image_path_with_same_class(self, image_path: str):
        stem = Path(image_path).stem
        cls = self.image_path_to_class_str_dict[stem]
        same = [p for p, c in self.image_path_to_class_str_dict.items() if c == cls]
        return [str(self.images_path / f"{p}{getattr(self, 'image_ext', '.jpg')}") for p in same]


class PlanesUtils(BaseUtils):
    """FGVC-Aircraft (all_utils/dataset_utils.py:180-224): images_{split}.txt lists the image
    ids; class string = '{manufacturer} {variant}'.  The dataset must already be on disk."""

    def __init__(self, split="train", root_path="data/FGVC-Aircraft/fgvc-aircraft-2013b/data", print_func=print,
                 image_ext=".jpg"):
        super().__init__(split, root_path, print_func=print_func)
        self.name = "planes"
        self.meta_class = "airplane"
        self.image_ext = image_ext
        self.images_path = Path(root_path) / "images"
        self.images_folder = self.root_path / "images"
        self.txt_file_path = self.root_path / f"images_{split}.txt"
        self.manufacturers_file_path = self.root_path / f"images_manufacturer_{split}.txt"
        self.variants_file_path = self.root_path / f"images_variant_{split}.txt"
        if not self.txt_file_path.exists():
            raise FileNotFoundError(f"{self.txt_file_path} not found: place FGVC-Aircraft under {self.root_path} "
                                    "(this build does not download datasets)")
        with open(self.txt_file_path, "r") as f:
            self.image_names = f.read().splitlines()
        self.original_images_paths = [str(self.images_folder / f"{n}{image_ext}") for n in self.image_names]
        self.print_func(f"Loaded {len(self.original_images_paths)} images for {self.name}")
        self.image_path_to_class_str_dict = self.get_image_stem_to_class_str_dict()

    def get_image_stem_to_class_str_dict(self):
        man = utils.load_data(self.manufacturers_file_path)
        var = utils.load_data(self.variants_file_path)
        return {i: f"{man[i]} {var[i]}" for i in man if i in var}

    def get_basic_prompt(self):
        return "a photo of an aircraft"


class SyntheticUtils(PlanesUtils):
    """Seeded synthetic stand-in with the FGVC-Aircraft layout (PNG sources)."""

    MANUFACTURERS = ["Boeing", "Airbus", "Cessna", "Embraer"]
    VARIANTS = ["707-320", "A320", "172", "ERJ 145", "747-400", "A380"]

    def __init__(self, split="train", root_path="data/synthetic-planes/data", print_func=print, n_images=16,
                 sizes=((512, 512),), seed=0):
        root = Path(root_path)
        if not (root / f"images_{split}.txt").exists():
            self._materialise(root, split, n_images, sizes, seed)
        super().__init__(split, root_path, print_func=print_func, image_ext=".png")
        self.name = "synthetic"

    @staticmethod
    def _materialise(root, split, n_images, sizes, seed):
        from PIL import Image

        from .synthetic import synthetic_image
        (root / "images").mkdir(parents=True, exist_ok=True)
        ids = [f"{1000000 + seed * 1000 + i:07d}" for i in range(n_images)]
        for i, image_id in enumerate(ids):
            h, w = sizes[i % len(sizes)]
            Image.fromarray(synthetic_image(h, w, seed * 1000 + i)).save(root / "images" / f"{image_id}.png")
        m, v = SyntheticUtils.MANUFACTURERS, SyntheticUtils.VARIANTS
        with open(root / f"images_{split}.txt", "w") as f:
            f.write("\n".join(ids) + "\n")
        with open(root / f"images_manufacturer_{split}.txt", "w") as f:
            f.write("".join(f"{image_id} {m[i % len(m)]}\n" for i, image_id in enumerate(ids)))
        with open(root / f"images_variant_{split}.txt", "w") as f:
            f.write("".join(f"{image_id} {v[i % len(v)]}\n" for i, image_id in enumerate(ids)))


DS_UTILS_DICT = {"planes": PlanesUtils, "synthetic": SyntheticUtils}

"""Per-dataset path / class bookkeeping touched by the generation loop (mirror of the thin
part of the reference's all_utils/dataset_utils.py: `original_images_paths`, `root_path`,
`meta_class`, `get_image_stem_to_class_str_dict`, `get_image_path_with_same_class`,
`get_basic_prompt`) for planes, cars, dtd, cub and compcars-parts.  The torchvision
downloads and `planes_biased` (ALIA's contextual-bias split) are out of scope; the baseline-classifier loader lives in
filters.py.

`SyntheticUtils` builds an on-disk dataset with the FGVC-Aircraft file layout from seeded
synthetic images, for the boxes that have no datasets (tests / bench / demo runs)."""
import glob
import os
from pathlib import Path

from . import utils

DATASETS_SUPPORTED = ["planes", "cars", "dtd", "compcars-parts", "cub", "synthetic"]
DATASETS_FILES = Path(__file__).parent / "datasets_files"      # split lists of the reference (fgvc/datasets_files), data


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

    def get_image_path_to_class_id_dict(self, split="train"):
        """image path -> integer label of the baseline classifier (the confidence filter looks the SOURCE image's label
        up here, all_utils/utils.py:316, :359).  Must number the classes like the dataset the classifier was trained on."""
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

    def get_image_path_to_class_id_dict(self, split="train"):
        """torchvision.datasets.FGVCAircraft(annotation_level="variant") numbering (fgvc/datasets/aircraft_dataset.py:19):
        class id = line number of the variant in data/variants.txt (file order); sorted variant names when that file is
        absent (the synthetic stand-in)."""
        variants = utils.load_data(self.root_path / f"images_variant_{split}.txt")
        vfile = self.root_path / "variants.txt"
        if vfile.exists():
            with open(vfile, "r") as f:
                classes = [ln.strip() for ln in f if ln.strip()]
        else:
            classes = sorted(set(variants.values()))
        idx = {c: i for i, c in enumerate(classes)}
        return {str(self.images_folder / f"{name}{self.image_ext}"): idx[v] for name, v in variants.items()}


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


def _read_val_list(name):
    """fgvc/datasets_files/{name}_val.txt of the reference: the validation images carved out of the official train split
    (shipped as data under saspa-aug_amd/datasets_files/)."""
    with open(DATASETS_FILES / f"{name}_val.txt", "r") as f:
        return {line.strip() for line in f}


class CarsUtils(BaseUtils):
    """Stanford Cars (all_utils/dataset_utils.py:227-298): cars_{train,test}/*.jpg, devkit/cars_meta.mat (class names),
    devkit/cars_{split}_annos.mat; "val" is a listed subset of the official train split.  Class string = the sub-model
    name ("Audi S4 Sedan 2012").  The image list is SORTED (the reference takes glob order, which depends on the file
    system; the plan must be identical on every rank -- SURVEY 8e)."""

    def __init__(self, split="train", root_path="data/stanford_cars/stanford_cars", print_func=print):
        super().__init__(split, root_path, print_func=print_func)
        self.name = "cars"
        self.meta_class = "car"
        assert split in ["train", "val", "test"]
        self.images_path = Path(root_path) / "cars_train"
        devkit = self.root_path / "devkit"
        self.meta_file_path = devkit / "cars_meta.mat"
        split_to_use = "train" if split == "val" else split
        self.annots_path = devkit / f"cars_{split_to_use}_annos.mat"
        self.images_folder = self.root_path / f"cars_{split_to_use}"
        if not self.images_folder.is_dir():
            raise FileNotFoundError(f"{self.images_folder} not found: place Stanford Cars under {self.root_path}")
        paths = sorted(glob.glob(f"{self.images_folder}/*.jpg"))
        if split in ("train", "val"):
            val = _read_val_list("cars")
            paths = [p for p in paths if (Path(p).name in val) == (split == "val")]
        self.original_images_paths = paths
        self.print_func(f"Loaded {len(paths)} images for cars, split {split}")
        self.image_path_to_class_str_dict = self.get_image_stem_to_class_str_dict()

    def get_image_stem_to_class_str_dict(self):
        import scipy.io as sio
        names = {i + 1: str(info[0]) for i, info in enumerate(sio.loadmat(self.meta_file_path)["class_names"][0])}
        out = {}
        for ann in sio.loadmat(self.annots_path)["annotations"][0]:
            class_id = int(ann[4][0][0])
            if class_id in names:
                out[Path(str(ann[-1][0])).stem] = names[class_id]
        return out

    def get_classes(self):
        return sorted(set(self.get_image_stem_to_class_str_dict().values()))

    def get_basic_prompt(self):
        return "a photo of a car"

    def get_image_path_to_class_id_dict(self, split="train"):
        """torchvision.datasets.StanfordCars numbering: annotation class - 1 (fgvc/datasets/car_dataset.py)."""
        import scipy.io as sio
        split_to_use = "train" if split == "val" else split
        ann = sio.loadmat(self.root_path / "devkit" / f"cars_{split_to_use}_annos.mat")["annotations"][0]
        folder = self.root_path / f"cars_{split_to_use}"
        return {str(folder / str(a[-1][0])): int(a[4][0][0]) - 1 for a in ann}


class _PathKeyedUtils(BaseUtils):
    """Datasets whose class table is keyed by the image PATH (dtd, cub, compcars-parts): the loop looks the sub-class up
    with `image_classes_dict[source_image_path]` (run_aug/run_aug.py:412-427) and same-class images are the keys."""

    def get_image_path_to_class_str_dict(self):
        raise NotImplementedError

    def get_image_stem_to_class_str_dict(self):
        # the generation loop asks for "the" class table through one name; for these datasets it is the path-keyed one
        # (run_aug/run_aug.py:700)
        return self.get_image_path_to_class_str_dict()

    def get_image_path_with_same_class(self, image_path: str):
        cls = self.image_path_to_class_str_dict[image_path]
        return [p for p, c in self.image_path_to_class_str_dict.items() if c == cls]


class DTDUtils(_PathKeyedUtils):
    """Describable Textures (all_utils/dataset_utils.py:302-338): images/<texture>/<name>.jpg, labels/{split}{partition}.txt
    lists "<texture>/<name>.jpg"; class string = the folder name.  DTD is generated from per-image BLIP captions
    (PROMPT_TYPE is forced to "captions", run_aug/run_aug.py:611-616)."""

    def __init__(self, split="train", partition=1, root_path="data/DTD/dtdataset/dtd", print_func=print):
        super().__init__(split, root_path, print_func=print_func)
        self.name = "dtd"
        self.meta_class = "texture"
        self.images_folder = self.root_path / "images"
        labels = self.root_path / "labels" / f"{split}{partition}.txt"
        if not labels.exists():
            raise FileNotFoundError(f"{labels} not found: place DTD under {self.root_path}")
        self.all_original_images_paths = sorted(glob.glob(f"{self.images_folder}/*/*.jpg"))
        with open(labels, "r") as f:
            self.original_images_paths = [str(self.images_folder / n) for n in f.read().splitlines()]
        self.print_func(f"Loaded {len(self.original_images_paths)} images for DTD split {split} partition {partition}")
        self.image_path_to_class_str_dict = self.get_image_path_to_class_str_dict()

    def get_classes(self):
        return sorted(os.listdir(self.images_folder))

    def get_image_path_to_class_str_dict(self):
        return {p: Path(p).parent.name for p in self.all_original_images_paths}

    def get_basic_prompt(self):
        return "a photo of a texture"

    def get_image_path_to_class_id_dict(self, split="train"):
        """torchvision.datasets.DTD numbering: index of the texture in the sorted class list; every image of the dataset
        (train + val + test of the partition, all_utils/dataset_utils.py:325-336)."""
        idx = {c: i for i, c in enumerate(sorted({Path(p).parent.name for p in self.all_original_images_paths}))}
        return {p: idx[Path(p).parent.name] for p in self.all_original_images_paths}


class CUBUtils(_PathKeyedUtils):
    """CUB-200-2011 (all_utils/dataset_utils.py:448-490 + fgvc/datasets/cub_dataset.py:41-78): images.txt /
    image_class_labels.txt / train_test_split.txt in file order; "train" = official training images minus the listed
    validation subset; class string = classes.txt name with the "NNN." prefix removed ("Black_footed_Albatross")."""

    def __init__(self, split="train", root_path="data/CUB/CUB_200_2011", print_func=print):
        super().__init__(split, root_path, print_func=print_func)
        self.name = "cub"
        self.meta_class = "bird"
        self.images_folder = self.root_path / "images"
        if not (self.root_path / "images.txt").exists():
            raise FileNotFoundError(f"{self.root_path / 'images.txt'} not found: place CUB_200_2011 under {self.root_path}")
        image_path = {}
        with open(self.root_path / "images.txt") as f:
            for line in f:
                image_id, rel = line.strip().split(" ")
                image_path[image_id] = str(Path(root_path) / "images" / rel)
        files = []
        with open(self.root_path / "train_test_split.txt") as f:
            for line in f:
                image_id, is_train = line.strip().split(" ")
                if (int(is_train) == 1) == (split in ("train", "val")):
                    files.append(image_path[image_id])
        if split in ("train", "val"):
            val = _read_val_list("cub")
            files = [p for p in files if (str(Path(*Path(p).parts[-2:])) in val) == (split == "val")]
        self.original_images_paths = files
        self.print_func(f"Loaded {len(files)} images for CUB")
        self.image_path_to_class_str_dict = self.get_image_path_to_class_str_dict()

    def get_image_path_to_class_str_dict(self):
        names = {}
        with open(self.root_path / "classes.txt") as f:
            for line in f:
                class_id, name = line.strip().split(" ", 1)
                names[int(class_id) - 1] = name.split(".")[1]
        return {p: names[int(Path(p).parent.name.split(".")[0]) - 1] for p in self.original_images_paths}

    def get_classes(self):
        return sorted(set(self.image_path_to_class_str_dict.values()))

    def get_basic_prompt(self):
        return "a photo of a bird"

    def get_image_path_to_class_id_dict(self, split="train"):
        """image_class_labels.txt label - 1 (fgvc/datasets/cub_dataset.py:47-51)."""
        label, path = {}, {}
        with open(self.root_path / "image_class_labels.txt") as f:
            for line in f:
                image_id, lb = line.strip().split(" ")
                label[image_id] = int(lb) - 1
        with open(self.root_path / "images.txt") as f:
            for line in f:
                image_id, rel = line.strip().split(" ")
                path[image_id] = str(self.root_path / "images" / rel)
        return {path[i]: label[i] for i in path}


class CompCarsPartsUtils(_PathKeyedUtils):
    """CompCars car-parts (all_utils/dataset_utils.py:342-444): part/<make id>/<model id>/<year>/<part id>/<name>.jpg,
    misc/make_model_name.mat; the train / test lists are CSVs of "path,label" (shipped as data); class string =
    "{make} {model}"; the prompt names the photographed part (get_basic_prompt(part))."""

    PART_TO_STRING = {"1": "Headlight", "2": "Taillight", "3": "Fog light", "4": "front"}

    def __init__(self, split="train", root_path="data/compcars", print_func=print):
        super().__init__(split, root_path, print_func=print_func)
        import scipy.io as sio
        self.name = "compcars-parts"
        self.meta_class = "car"
        assert split in ["train", "val", "test"]
        split_to_use = "train" if split == "val" else split
        self.images_folder = self.root_path / "part"
        mat_path = self.root_path / "misc/make_model_name.mat"
        if not mat_path.exists():
            raise FileNotFoundError(f"{mat_path} not found: place CompCars under {self.root_path}")
        mat = sio.loadmat(mat_path)

        def names(key):
            return [None if len(x[0]) == 0 else x[0].item() for x in mat[key]]
        makes, models = names("make_names"), names("model_names")
        self.full_folder_path_to_make_model = {}
        for folder in sorted(glob.glob(f"{self.images_folder}/*/*")):
            mk, md = makes[int(folder.split("/")[-2]) - 1], models[int(folder.split("/")[-1]) - 1]
            self.full_folder_path_to_make_model[folder] = f"{'<NA>' if mk is None else mk} {'<NA>' if md is None else md}"

        def read_csv(name):
            rows = []
            with open(DATASETS_FILES / "compcars-parts" / name) as f:
                for line in f:
                    path, label = line.strip().split(",")
                    rows.append((str(Path("data/compcars/part") / path), label))
            return rows
        split_rows, all_rows = read_csv(f"{split_to_use}.csv"), read_csv("train_and_test.csv")
        self.original_images_paths = [p for p, _ in split_rows]
        self.all_original_images_paths = [p for p, _ in all_rows]
        if split in ("train", "val"):
            val = _read_val_list("compcars_parts")
            self.original_images_paths = [p for p in self.original_images_paths if (p in val) == (split == "val")]
        self.all_classes = sorted({label for _, label in all_rows})
        self.all_classes_as_strings = sorted({self._make_model(p) for p in self.original_images_paths})
        self.part_to_string = dict(self.PART_TO_STRING)
        self.print_func(f"Loaded {len(self.original_images_paths)} images for compcars dataset split {split}")
        self.image_path_to_class_str_dict = self.get_image_path_to_class_str_dict()

    def _make_model(self, image_path):
        return self.full_folder_path_to_make_model[str(Path(image_path).parent.parent.parent)]

    def get_classes(self):
        return self.all_classes_as_strings

    def get_image_path_to_class_str_dict(self):
        return {p: self._make_model(p) for p in self.all_original_images_paths}

    def get_image_path_to_class_id_dict(self, split="train"):
        """Labels of the split CSV numbered by their sorted order (all_utils/dataset_utils.py:413-430)."""
        rows = []
        with open(DATASETS_FILES / "compcars-parts" / f"{split}.csv") as f:
            for line in f:
                pth, lb = line.strip().split(",")
                rows.append((str(Path("data/compcars/part") / pth), lb))
        idx = {lb: i for i, lb in enumerate(sorted({lb for _, lb in rows}))}
        return {pth: idx[lb] for pth, lb in rows}

    def get_basic_prompt(self, part: str = None):
        return f"close up of the {self.part_to_string[str(part)]} of a" if part else "close up of a car"

    def get_image_path_with_same_class(self, image_path: str):
        cls, part = self.image_path_to_class_str_dict[image_path], image_path.split("/")[-2]
        return [p for p, c in self.image_path_to_class_str_dict.items() if c == cls and p.split("/")[-2] == part]


DS_UTILS_DICT = {"planes": PlanesUtils, "synthetic": SyntheticUtils, "cars": CarsUtils, "dtd": DTDUtils,
                 "compcars-parts": CompCarsPartsUtils, "cub": CUBUtils}

"""Host-side helpers of the generation path, mirroring the reference's `all_utils/utils.py`
interface for this path (same names, argument meaning and outputs):

  set_seed                      all_utils/utils.py:32-36
  HWC3 / resize_image           all_utils/utils.py:39-79
  generate_canny (+ preprocess_canny, CannyDetector -> HIP kernel)   all_utils/utils.py:81-109
  get_aug_json_path             all_utils/utils.py:194-218
  create_json_of_image_name_to_augmented_images_paths   all_utils/utils.py:221-465 (PNG integrity
        sweep, stem matching, JSON layout, and the semantic / model-confidence filters, which run
        on the gfx950 kernels: saspa_aug_amd/filters.py; LPIPS / per-class CLIP / ALIA filters are
        baseline branches and raise NotImplementedError)
  check_folder_of_images_with_pil   all_utils/utils.py:681-703
  init_logging                  all_utils/utils.py:593-612

The Canny arithmetic runs in the gfx950 kernel (saspa_canny); everything else here is
file / string bookkeeping.

Mirrored text, declared: about 45 lines of this file follow the reference's wording closely because they ARE the contract
(a byte-exact file name, array shape or RNG call sequence that `fgvc/train.py` and the golden tests check): `set_seed`,
`HWC3` (12 lines), the target-size arithmetic of `resize_image` (7 lines; the resampling itself is the HIP kernel), the
4-line body of `generate_canny`, and `get_aug_json_path` (the tag concatenation that produces e.g.
`semantic_filtering-model_confidence_based_filtering_top_10_classes-aug.json`).  Everything else is this build's own."""
import datetime
import json
import logging
import os
import random
from pathlib import Path

import numpy as np
import torch
from PIL import Image

MAX_FILE_NAME_LENGTH = 40   # must equal run_aug.MAX_FILENAME_LENGTH (all_utils/utils.py:342)
SUBSTRINGS_TO_EXCLUDE = ["_source.", "_style.", "_target.", "_control.", "_original.", "_subject.", "subject_"]


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def HWC3(x):
    assert x.dtype == np.uint8
    if x.ndim == 2:
        x = x[:, :, None]
    assert x.ndim == 3
    H, W, C = x.shape
    assert C == 1 or C == 3 or C == 4
    if C == 3:
        return x
    if C == 1:
        return np.concatenate([x, x, x], axis=2)
    color = x[:, :, 0:3].astype(np.float32)
    alpha = x[:, :, 3:4].astype(np.float32) / 255.0
    y = color * alpha + 255.0 * (1.0 - alpha)
    return y.clip(0, 255).astype(np.uint8)


def resize_target_size(h, w, smaller_side_res):
    """The (H, W) `resize_image` resizes to: smaller side -> res, area capped at 1.2 MP, both
    sides rounded to multiples of 64; also returns the final scale k (k > 1 -> upscaling)."""
    MAX_RES_SIZE = 1200000
    H, W = float(h), float(w)
    k = float(smaller_side_res) / min(H, W)
    H *= k
    W *= k
    if H * W > MAX_RES_SIZE:
        k = np.sqrt(MAX_RES_SIZE / (H * W))
        H *= k
        W *= k
    return int(np.round(H / 64.0)) * 64, int(np.round(W / 64.0)) * 64, k


def resize_image(input_image, smaller_side_res):
    """all_utils/utils.py:58-79: target size as the reference computes it, cv2.resize semantics (INTER_LANCZOS4 when the
    final scale k > 1, INTER_AREA otherwise) through the gfx950 kernels (saspa_resize_taps_u8 / saspa_resize_area_u8).
    Identity (no device needed) when the image already has the target size.  numpy u8 [H,W,3] in and out."""
    H, W, _ = input_image.shape
    th, tw, k = resize_target_size(H, W, smaller_side_res)
    if (th, tw) == (H, W):
        return input_image
    if not torch.cuda.is_available():
        raise RuntimeError("resize_image runs on the MI355X only (no CPU path)")
    from . import imageproc, ops
    src = ops.h2d(torch.from_numpy(np.ascontiguousarray(input_image))[None], torch.device("cuda", torch.cuda.current_device()))
    return imageproc.cv_resize_u8(src, th, tw, "lanczos4" if k > 1 else "area")[0].cpu().numpy()


class CannyDetector:
    """cv2.Canny(img, low, high) -> the gfx950 kernel (integer exact, see csrc/saspa_canny.hip)."""

    def __call__(self, img, low_threshold, high_threshold):
        from . import ops
        if not torch.cuda.is_available():
            raise RuntimeError("CannyDetector runs on the MI355X only (no CPU path)")
        t = torch.from_numpy(np.ascontiguousarray(img))[None].to("cuda")
        return ops.canny(t, low_threshold, high_threshold)[0, :, :, 0].cpu().numpy()


apply_canny = CannyDetector()


def preprocess_canny(input_image, image_resolution, low_threshold, high_threshold):
    image = resize_image(HWC3(input_image), image_resolution)
    control_image = apply_canny(image, low_threshold, high_threshold)
    control_image = HWC3(control_image)
    return Image.fromarray(control_image)


def generate_canny(cond_image_input, low_threshold, high_threshold, image_resolution):
    cond_image_input = np.array(cond_image_input).astype(np.uint8)
    return preprocess_canny(cond_image_input, image_resolution, low_threshold=low_threshold, high_threshold=high_threshold)


def get_aug_json_path(augmented_image_folder_path, lpips_min=None, lpips_max=None, clip_filtering=False,
                      clip_filtering_discount=1, semantic_filtering=False, model_confidence_based_filtering=False,
                      conf_top_k: int = 10, filter_confidence_higher_than: int = None, alia_conf_filtering=False):
    json_name = ""
    if lpips_min:
        json_name += f"lpips_min_{lpips_min}-"
    if lpips_max:
        json_name += f"lpips_max_{lpips_max}-"
    if clip_filtering:
        json_name += f"clip_filtering_{clip_filtering}_discount_{clip_filtering_discount}-"
    if semantic_filtering:
        json_name += "semantic_filtering-"
    if model_confidence_based_filtering:
        json_name += f"model_confidence_based_filtering_top_{conf_top_k}_classes-"
        if filter_confidence_higher_than:
            json_name += f"filter_confidence_higher_than_{filter_confidence_higher_than}-"
    if alia_conf_filtering:
        json_name += "alia_conf_filtering-"
    json_name += "aug.json"
    return str(Path(augmented_image_folder_path).parent / json_name)


def check_folder_of_images_with_pil(folder, max_delete=20, substrings_to_exclude=None):
    """Integrity sweep over the generated files before they are listed in the JSON (interface of
    all_utils/utils.py:681-703): every file whose name carries none of `substrings_to_exclude` must pass PIL's
    structural check; broken ones (a writer killed mid-file) are removed so training never opens them.  Stops
    after `max_delete` removals -- more than that means the run itself is broken, not a few files."""
    skip = tuple(substrings_to_exclude or ())
    removed = 0
    with os.scandir(folder) as entries:
        candidates = sorted(e.path for e in entries if e.is_file() and not any(t in e.name for t in skip))
    for path in candidates:
        if removed >= max_delete:
            break
        try:
            with Image.open(path) as im:
                im.verify()
        except Exception as err:            # PIL raises several unrelated types for truncated / non-image files
            logging.info(f"image {path} is corrupted ({type(err).__name__}), deleting")
            os.remove(path)
            removed += 1
    logging.info(f"Finished checking folder {folder} with PIL, deleted {removed} images")
    return removed


def match_augmented_images(original_images_paths, all_file_names, augmented_image_folder_path):
    """{Path(orig).name: [str(Path(folder)/png), ...]} for EVERY original image (empty lists
    kept), matching by `stem[:40] in png_name` in directory-listing order
    (all_utils/utils.py:343-354, :437)."""
    names = [f for f in all_file_names if not any(s in f for s in SUBSTRINGS_TO_EXCLUDE)]
    out = {}
    for image_path in original_images_paths:
        image_name = Path(image_path).name
        stem = Path(image_name).stem[:MAX_FILE_NAME_LENGTH]
        out[image_name] = [str(Path(augmented_image_folder_path) / n) for n in names if stem in n]
    return out


def create_json_of_image_name_to_augmented_images_paths(dataset, augmented_image_folder_path, lpips_min=None, lpips_max=None,
                                                        resize=(256, 256), clip_filtering=False, clip_filtering_discount=1,
                                                        semantic_filtering=False, model_confidence_based_filtering=False,
                                                        conf_top_k: int = 10, filter_confidence_higher_than: int = None,
                                                        init_log=True, alia_conf_filtering=False, original_images_paths=None,
                                                        min_files=10, filter_models=None, weights_dir=None, device=None):
    """`dataset` may be a dataset name (resolved through dataset_utils.DS_UTILS_DICT) or any
    object with `.original_images_paths`."""
    assert not (clip_filtering and model_confidence_based_filtering)
    if any([lpips_min, lpips_max, clip_filtering, alia_conf_filtering]):
        raise NotImplementedError("LPIPS / CLIP-per-class / ALIA filters are baseline branches (out of scope, SURVEY 2 row 7)")
    if not str(augmented_image_folder_path).endswith("/images"):
        augmented_image_folder_path = str(Path(augmented_image_folder_path) / "images")
    json_path = get_aug_json_path(augmented_image_folder_path, lpips_min, lpips_max, clip_filtering, clip_filtering_discount,
                                  semantic_filtering, model_confidence_based_filtering, conf_top_k,
                                  filter_confidence_higher_than, alia_conf_filtering)
    if init_log:
        init_logging(logdir=None, logfile=json_path.replace(".json", ".log"))
    logging.info(f"json_path = {json_path}")
    check_folder_of_images_with_pil(augmented_image_folder_path, max_delete=50, substrings_to_exclude=SUBSTRINGS_TO_EXCLUDE)
    if original_images_paths is None:
        if isinstance(dataset, str):
            from . import dataset_utils
            dataset = dataset_utils.DS_UTILS_DICT[dataset](print_func=logging.info)
        original_images_paths = dataset.original_images_paths
    if len(list(Path(augmented_image_folder_path).glob("*.*"))) < min_files:
        raise FileNotFoundError(f"augmented_image_folder_path = {augmented_image_folder_path} doesn't exist or has less "
                                f"than {min_files} images")
    mapping = match_augmented_images(original_images_paths, os.listdir(augmented_image_folder_path), augmented_image_folder_path)
    if semantic_filtering or model_confidence_based_filtering:
        # the filter stage: CLIP-RN50 semantic filter + baseline-classifier top-k filter on the gfx950 kernels
        from . import filters
        if isinstance(dataset, str):
            from . import dataset_utils
            dataset = dataset_utils.DS_UTILS_DICT[dataset](print_func=logging.info)
        if filter_models is None:
            if device is None:
                device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
            if device is None:
                raise RuntimeError("the semantic / model-confidence filters run on the MI355X only (no CPU path); pass "
                                   "semantic_filtering=0, model_confidence_based_filtering=0 to write the unfiltered aug.json")
            filter_models = filters.build_filters(dataset, device, bool(semantic_filtering), bool(model_confidence_based_filtering),
                                                  weights_dir, conf_top_k)
        sem, conf = filter_models
        if filter_confidence_higher_than:
            raise NotImplementedError("filter_confidence_higher_than is an ablation knob of the reference (unused by run_aug)")
        mapping, counters = filters.apply_filters(mapping, original_images_paths, dataset, device, sem if semantic_filtering else None,
                                                  conf if model_confidence_based_filtering else None)
        if semantic_filtering:
            logging.info(f"For filter = semantic_filtering, filtered {counters['semantic']} images")
        if model_confidence_based_filtering:
            logging.info(f"For filter = not_in_top_{conf_top_k}, filtered {counters['not_in_top_k']} images")
    Path(json_path).parent.mkdir(parents=True, exist_ok=True)
    with open(json_path, "w") as f:
        json.dump(mapping, f)
    logging.info(f"Finished creating json of image name to augmented images paths in: \n{json_path}")
    counts = {}
    for v in mapping.values():
        counts[len(v)] = counts.get(len(v), 0) + 1
    logging.info(f"dict_num_augmentations_per_image = {dict(sorted(counts.items()))}")
    return json_path


def _log_file_for(logdir, logfile, stamp):
    if logdir:
        os.makedirs(logdir, exist_ok=True)
        return os.path.join(logdir, f"{stamp}_log.log")
    target = Path(logfile)
    target.parent.mkdir(parents=True, exist_ok=True)
    return str(target.with_name(f"{target.stem}_{stamp}{target.suffix}"))


def init_logging(logdir=None, logfile=None, return_logger=False):
    """Root logger to the console and to a time-stamped file: `{logdir}/{stamp}_log.log`, or `logfile` with the stamp
    inserted before its suffix (interface of all_utils/utils.py:593-612; called once per process, so the file handler is
    replaced rather than stacked when a second stage -- the JSON writer -- logs next to the first)."""
    if not (logdir or logfile):
        raise AssertionError("logdir or logfile must be provided")
    stamp = datetime.datetime.now().strftime("%Y_%m%d_%H%M_%S")
    fmt = "%(asctime)s %(levelname)s %(message)s"
    root = logging.getLogger()
    logging.basicConfig(format=fmt, level=logging.INFO)
    root.setLevel(logging.INFO)
    for h in [h for h in root.handlers if getattr(h, "_saspa_file", False)]:
        root.removeHandler(h)
        h.close()
    handler = logging.FileHandler(_log_file_for(logdir, logfile, stamp), mode="w")
    handler.setFormatter(logging.Formatter(fmt))
    handler._saspa_file = True
    root.addHandler(handler)
    return root if return_logger else logdir


def load_data(file_path):
    """FGVC-Aircraft annotation file -> {image id: rest of the line} ("1025794 Boeing" / "0734043 707-320";
    the value may itself contain spaces).  Interface of all_utils/utils.py:615-621."""
    with open(file_path, "r") as f:
        pairs = (line.rstrip("\n").strip().partition(" ") for line in f)
        return {image_id: info for image_id, _, info in pairs if image_id}

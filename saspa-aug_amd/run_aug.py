"""SaSPA augmentation generation -- the host side of the hot path, mirroring the reference's
`run_aug/run_aug.py` (constants :47-72, init_pipeline :128-230, pass_thorugh_pipe :233-279,
main :282-504, config/output-tree block :507-733) behind the same switches
(BASE_MODEL / CONTROLNET / NUM_PER_IMAGE / SEED / PROMPT_TYPE ...) and the same output tree /
file names / JSON, so `fgvc/train.py` consumes the results unchanged.

What is different by design (MI355X-first; results per work item are unchanged):
  * the reference generates one variant per `pipe()` call; here a cheap host-only PLANNING
    pass replays the reference's host RNG (numpy / random / the CPU noise generator, in the
    reference's exact order) into a manifest of work items, which are bucketed by image size,
    batched (B images = 2B CFG samples per launch sequence) and sharded across ranks;
  * each item's initial noise is the slice of the single sequential CPU noise stream the
    reference would have drawn for it (offset = sum of the preceding latent sizes), so
    batching and sharding do not change any image;
  * Canny runs once per source image on the GPU (the reference recomputes it per variant);
  * PNG encoding is asynchronous (thread pool) so it does not serialise the GPU;
  * one process per GPU (`torchrun`), no model parallelism; the only collective is one gather
    of the per-item status vector to rank 0, which then writes the JSON manifest."""
import logging
import os
import random
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from . import dataset_utils, utils
from .prompts_engineering import ARTISTIC_PROMPTS, IMAGE_VARIATIONS_PROMPTS

NEGATIVE_PROMPT = ("over-exposure, under-exposure, saturated, duplicate, out of frame, lowres, cropped, worst quality, "
                   "low quality, jpeg artifacts, morbid, mutilated, out of frame, ugly, bad anatomy, bad proportions, "
                   "deformed, blurry, duplicate")                       # run_aug/run_aug.py:47
MAX_FILENAME_LENGTH = 40
MAX_PROMPT_LENGTH = 150

BASE_MODEL_DICT = {
    "sd_v1.5": "runwayml/stable-diffusion-v1-5",
    "sd_v2.1": "stabilityai/stable-diffusion-2-1-base",
    "sd_xl": ["stabilityai/stable-diffusion-xl-base-1.0", "stabilityai/stable-diffusion-xl-refiner-1.0"],
    "sd_xl-turbo": "stabilityai/sdxl-turbo",
    "blip_diffusion": "Salesforce/blipdiffusion",
    "blip_diffusion-controlnet": "Salesforce/blipdiffusion-controlnet",
    "ip2p": "timbrooks/instruct-pix2pix",
}
CONTROLNET_DICT_SD = {"canny": "lllyasviel/control_v11p_sd15_canny", "hed": "lllyasviel/sd-controlnet-hed"}
CONTROLNET_DICT_SD_XL = {"canny": "diffusers/controlnet-canny-sdxl-1.0"}


def _save_png(arr, path):
    Image.fromarray(arr).save(path)


class _PngWriters:
    """PNG encoding off the launch thread.  Pillow's encoder holds the GIL (4 encoding threads cut the main thread's
    Python throughput to 27 %: measured), and the main thread is what enqueues the GPU work, so the writers are separate
    PROCESSES: `png_worker.py` children started with subprocess (no fork of this -- large, GPU-holding -- process image,
    nothing but numpy / Pillow in the children), each fed through its stdin by a feeder thread (pipe writes release the
    GIL).  SASPA_PNG_PROCS=0 selects the old in-process thread pool."""

    def __init__(self, workers=4):
        import queue
        import subprocess
        import sys
        import threading
        n = int(os.environ.get("SASPA_PNG_PROCS", workers))
        self.procs, self.queues, self.feeders, self.next = [], [], [], 0
        self.threads, self.pending = None, []
        self.submitted, self.max_depth = 0, 0        # images handed over / deepest backlog seen at a submit (diagnostics)
        if n <= 0:
            self.threads = ThreadPoolExecutor(max_workers=workers)
            return
        script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "png_worker.py")

        def feed(proc, q):
            while True:
                item = q.get()
                if item is None:
                    break
                header, data = item
                proc.stdin.write(header)
                proc.stdin.write(data)
            proc.stdin.close()

        for _ in range(n):
            proc = subprocess.Popen([sys.executable, script], stdin=subprocess.PIPE)
            q = queue.Queue()
            t = threading.Thread(target=feed, args=(proc, q), daemon=True)
            t.start()
            self.procs.append(proc)
            self.queues.append(q)
            self.feeders.append(t)

    def submit(self, arr, path):
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        if self.threads is not None:
            self.pending.append(self.threads.submit(_save_png, arr, str(path)))
            return
        h, w = arr.shape[:2]
        c = arr.shape[2] if arr.ndim == 3 else 1
        if "\n" in str(path):
            raise ValueError("newline in an output path")
        self.queues[self.next].put((f"{h} {w} {c} {path}\n".encode("utf-8"), arr.tobytes()))
        self.next = (self.next + 1) % len(self.queues)
        self.submitted += 1
        self.max_depth = max(self.max_depth, sum(q.qsize() for q in self.queues))

    def close(self):
        if self.threads is not None:
            for f in self.pending:
                f.result()
            self.threads.shutdown()
            return
        for q in self.queues:
            q.put(None)
        for t in self.feeders:
            t.join()
        codes = [p.wait() for p in self.procs]
        if any(codes):
            raise RuntimeError(f"PNG writer processes exited with {codes}")


@dataclass
class Settings:
    """The module-level constants of the reference's `__main__` block (run_aug/run_aug.py:513-556)."""
    DEBUG: int = 0
    SPECIFIC_FILE_STRs: list = None
    DEVICE: str = "cuda:0"
    version: str = "v1"
    DATASET: str = "planes"
    BASE_MODEL: str = "sd_v1.5"
    CONTROLNET: str = "canny"
    SDEDIT: int = 0
    NUM_PER_IMAGE: int = 2
    SEED: int = 1
    PROMPT_TYPE: str = "gpt-meta_class"
    PROMPT_WITH_SUB_CLASS: bool = True
    USE_ARTISTIC_PROMPTS: bool = True
    ARTISTIC_PROMPTS_PROB: float = 0.5
    USE_CAMERA_VARIATIONS_PROMPTS: bool = False
    CAMERA_VAIRATIONS_PROB: float = 0.5
    RESOLUTION: int = 512
    GUIDANCE_SCALE: float = 7.5
    NUM_INFERENCE_STEPS: int = 30
    SDEDIT_STRENGTH: float = 0.85
    LOW_THRESHOLD_CANNY: int = 120
    HIGH_THRESHOLD_CANNY: int = 200
    CONTROLNET_CONDITIONING_SCALE: float = 0.75
    SEMANTIC_FILTERING: int = 1
    MODEL_CONFIDENCE_BASED_FILTERING: int = 1
    STYLE_IMG_FROM_DIFF_IMG: bool = True   # blip_diffusion: subject image = another image of the same class (:548)
    # additions of this build
    BATCH_SIZE: int = 8
    PRECISION: str = "bf16"            # "bf16" (production) | "fp32" (parity mode)
    WEIGHTS_DIR: str = None            # local diffusers-format checkpoints; None -> synthetic weights
    PROMPTS_FILE: str = None
    DATASET_KWARGS: dict = field(default_factory=dict)
    MAX_BATCHES: int = 0               # > 0: stop this rank after that many batches (rehearsals / diagnostics; the rest stays status 0)


@dataclass
class WorkItem:
    order: int            # position in the reference's loop order
    index: int            # source image index
    source_path: str
    image_stem: str
    i: int                # variant number
    prompt: str
    output_path: str
    height: int
    width: int
    noise_offset: int = -1   # element offset into the sequential fp16/fp32 CPU noise stream
    subject_path: str = None # BLIP-Diffusion: the same-class image whose subject is injected (run_aug/run_aug.py:446)
    skip: bool = False
    status: int = 0       # 0 skipped (exists), 1 generated, -1 failed


# ------------------------------------------------------------------------------------------
# configuration -> paths (run_aug/run_aug.py:668-692)
# ------------------------------------------------------------------------------------------
def prompt_str_for(s: Settings):
    p = s.PROMPT_TYPE
    if s.PROMPT_WITH_SUB_CLASS:
        p += "_prompt_w_sub_class"
    if s.USE_ARTISTIC_PROMPTS:
        p += f"_artistic_prompts_p_{s.ARTISTIC_PROMPTS_PROB}"
    if s.USE_CAMERA_VARIATIONS_PROMPTS:
        p += f"_camera_variations_p_{s.CAMERA_VAIRATIONS_PROB}"
    if "blip_diffusion" in s.BASE_MODEL and s.STYLE_IMG_FROM_DIFF_IMG:
        p += "_style_img_from_diff_img"
    return p


def output_folder_for(s: Settings, root_path):
    base_model_folder = f"regular/{s.BASE_MODEL}"
    if s.SDEDIT:
        base_model_folder += f"-SDEdit_strength_{s.SDEDIT_STRENGTH}"
    if s.CONTROLNET:
        base_model_folder = base_model_folder.replace("regular/", "controlnet/")
    return f"{root_path}/aug_data/{base_model_folder}/{s.CONTROLNET}/{prompt_str_for(s)}_seed_{s.SEED}/images"


def read_prompts(prompts_file):
    with open(prompts_file, "r") as f:
        prompts = [p.strip()[:MAX_PROMPT_LENGTH] for p in f.readlines()]
    return prompts


def read_prompts_from_json(json_file, dataset_name=None, per_class=False):
    """prompts_engineering/blip_utils.py:14-27: a {class: [prompts]} JSON -> the dict itself (per_class) or every class's
    prompts in file order (PROMPT_TYPE "txt2sentence-per_class" / "txt2sentence", run_aug/run_aug.py:331-339); truncated to
    MAX_PROMPT_LENGTH like the reference does after reading."""
    import json as _json
    with open(json_file, "r") as f:
        d = _json.load(f)
    if per_class:
        return {k: [q[:MAX_PROMPT_LENGTH] for q in v] for k, v in d.items()}
    out = []
    for v in d.values():
        out += v
    return [q[:MAX_PROMPT_LENGTH] for q in out]


# ------------------------------------------------------------------------------------------
# pipeline construction / call (run_aug/run_aug.py:128-279)
# ------------------------------------------------------------------------------------------
def init_pipeline(base_model, controlnet, SDEdit, use_compile=False, sampler="ddim", weights_dir=None, cfgs=None,
                  state_dicts=None):
    """(base_model, controlnet, SDEdit) -> pipeline object.  Implemented: sd_v1.5 + canny (the
    SaSPA configuration for planes and the BASELINE metric).  `use_compile` is accepted and
    ignored: the reference's torch.compile(reduce-overhead) has no counterpart, the kernels are
    launched directly."""
    from .config import BLIP_DIFFUSION, SD15, SDXL_TURBO
    from .pipeline import (BlipDiffusionControlNetPipeline, StableDiffusionControlNetPipeline,
                           StableDiffusionXLControlNetPipeline)
    from .scheduler import DDIMScheduler
    assert base_model in BASE_MODEL_DICT.keys()
    assert controlnet in CONTROLNET_DICT_SD.keys() or controlnet in CONTROLNET_DICT_SD_XL.keys() or controlnet is None
    assert sampler in ["ddim", "unipcmultistep"]
    if SDEdit and sampler == "unipcmultistep" and "blip_diffusion" not in base_model:
        # the reference allows it (run_aug/run_aug.py:216-221 applies to both img2img pipelines) but never selects it (main()
        # calls init_pipeline without `sampler`, :323); here the UniPC multistep history cannot start mid-schedule, so refuse
        # at start-up instead of on the first batch, after the weights were loaded and the work planned
        raise NotImplementedError("SDEdit (img2img) with sampler='unipcmultistep' is not built: use sampler='ddim' (the reference's default)")
    if base_model == "blip_diffusion" and controlnet == "canny" and not SDEdit:
        # run_aug/run_aug.py:178-181, :211: BlipDiffusionControlNetPipeline from Salesforce/blipdiffusion-controlnet; the
        # checkpoint's PNDM scheduler is KEPT (:217 switches only non-BLIP pipelines to DDIM / UniPC)
        cfgs = cfgs or BLIP_DIFFUSION
        if state_dicts is not None:
            return BlipDiffusionControlNetPipeline(state_dicts, cfgs)
        if weights_dir:
            return BlipDiffusionControlNetPipeline.from_pretrained(os.path.join(weights_dir, BASE_MODEL_DICT["blip_diffusion-controlnet"]), cfgs)
        logging.info("no WEIGHTS_DIR given: using architecture-exact SYNTHETIC weights (no checkpoint available offline)")
        return BlipDiffusionControlNetPipeline.from_synthetic(cfgs, seed=0)
    def _scheduler_for(pipe):
        # run_aug/run_aug.py:216-221 (and :223-228 for sd_xl-turbo): the non-BLIP pipelines get UniPC or DDIM built from the
        # checkpoint's scheduler config
        from .scheduler import UniPCMultistepScheduler
        cls = UniPCMultistepScheduler if sampler == "unipcmultistep" else DDIMScheduler
        return cls.from_config(pipe.scheduler.config)
    if base_model == "sd_xl-turbo" and controlnet == "canny" and not SDEdit:
        # run_aug/run_aug.py:185-201: ControlNetModel(diffusers/controlnet-canny-sdxl-1.0) + sdxl-vae-fp16-fix +
        # StableDiffusionXLControlNetPipeline(stabilityai/sdxl-turbo); :217-228 DDIM from the pipeline's scheduler
        # config (twice), upcast_vae()
        cfgs = cfgs or SDXL_TURBO
        if state_dicts is not None:
            pipe = StableDiffusionXLControlNetPipeline(state_dicts, cfgs)
        elif weights_dir:
            vae_dir = os.path.join(weights_dir, "madebyollin/sdxl-vae-fp16-fix")
            pipe = StableDiffusionXLControlNetPipeline.from_pretrained(
                os.path.join(weights_dir, BASE_MODEL_DICT[base_model]), os.path.join(weights_dir, CONTROLNET_DICT_SD_XL[controlnet]),
                vae_dir if os.path.isdir(vae_dir) else None, cfgs)
        else:
            logging.info("no WEIGHTS_DIR given: using architecture-exact SYNTHETIC weights (no checkpoint available offline)")
            pipe = StableDiffusionXLControlNetPipeline.from_synthetic(cfgs, seed=0)
        pipe.scheduler = _scheduler_for(pipe)        # DDIM or UniPC, "trailing" spacing inherited from the sdxl-turbo config
        pipe.upcast_vae()
        return pipe
    if base_model == "sd_v1.5" and controlnet is None and SDEdit:
        # run_aug/run_aug.py:163-165: StableDiffusionImg2ImgPipeline -- the Real-Guidance form (CONTROLNET = None, SDEDIT = 1;
        # run_aug/run_aug_real_guidance.py:520-523), scheduler switched like every non-BLIP pipeline (:216-221)
        from .pipeline import StableDiffusionImg2ImgPipeline
        cfgs = cfgs or SD15
        if state_dicts is not None:
            pipe = StableDiffusionImg2ImgPipeline(state_dicts, cfgs)
        elif weights_dir:
            pipe = StableDiffusionImg2ImgPipeline.from_pretrained(os.path.join(weights_dir, BASE_MODEL_DICT[base_model]), cfgs)
        else:
            logging.info("no WEIGHTS_DIR given: using architecture-exact SYNTHETIC weights (no checkpoint available offline)")
            pipe = StableDiffusionImg2ImgPipeline.from_synthetic(cfgs, seed=0)
        pipe.scheduler = _scheduler_for(pipe)
        return pipe
    if base_model != "sd_v1.5" or controlnet not in CONTROLNET_DICT_SD:
        raise NotImplementedError(
            f"({base_model}, {controlnet}, SDEdit={SDEdit}): sd_v1.5 (text-to-image and SDEdit img2img with the canny or HED "
            "ControlNet; ControlNet-free SDEdit img2img), blip_diffusion and sd_xl-turbo with the canny ControlNet are built; "
            "ip2p / blip_diffusion-edit / sd_v2.1 / sd_xl and the ControlNet-free text-to-image pipelines are baseline branches")
    cfgs = cfgs or SD15
    if SDEdit:                                  # run_aug/run_aug.py:203-206: StableDiffusionControlNetImg2ImgPipeline
        from .pipeline import StableDiffusionControlNetImg2ImgPipeline as _Cls
    else:
        _Cls = StableDiffusionControlNetPipeline
    if state_dicts is not None:
        pipe = _Cls(state_dicts, cfgs)
    elif weights_dir:
        pipe = _Cls.from_pretrained(
            os.path.join(weights_dir, BASE_MODEL_DICT[base_model]), os.path.join(weights_dir, CONTROLNET_DICT_SD[controlnet]), cfgs)
    else:
        logging.info("no WEIGHTS_DIR given: using architecture-exact SYNTHETIC weights (no checkpoint available offline)")
        pipe = _Cls.from_synthetic(cfgs, seed=0)
    pipe.scheduler = _scheduler_for(pipe)
    return pipe


def pass_thorugh_pipe(base_model, pipe, prompt, orig_img, SDEdit, SDEdit_strength, num_inference_steps, generator,
                      guidance_scale, control_cond_scale, negative_prompt=NEGATIVE_PROMPT, control_image=None,
                      blip_src_category=None, blip_target_category=None):
    """Single-variant call form of the reference (name kept, typo included)."""
    pipe_args = {"prompt": str(prompt), "num_inference_steps": num_inference_steps, "generator": generator,
                 "guidance_scale": guidance_scale, "negative_prompt": negative_prompt}
    if "ip2p" in base_model or base_model == "blip_diffusion-edit" or (control_image is None and not SDEdit):
        raise NotImplementedError("built call forms: ControlNet text-to-image / SDEdit img2img / BLIP-Diffusion, and the "
                                  "ControlNet-free SDEdit img2img (Real-Guidance)")
    if "blip_diffusion" in base_model:                     # run_aug/run_aug.py:243-250
        pipe_args["reference_image"] = orig_img
        pipe_args["source_subject_category"] = blip_src_category
        pipe_args["target_subject_category"] = blip_target_category
        pipe_args["height"] = orig_img.size[1]
        pipe_args["width"] = orig_img.size[0]
        pipe_args["neg_prompt"] = NEGATIVE_PROMPT
        del pipe_args["negative_prompt"]
    if control_image is not None:
        if SDEdit:                                         # :252-255 the img2img pipelines name the control image differently
            pipe_args["control_image"] = control_image
            pipe_args["controlnet_conditioning_scale"] = control_cond_scale
        elif "blip_diffusion" in base_model:               # :262-265 -- note: no conditioning scale is passed
            pipe_args["condtioning_image"] = control_image
            pipe_args["height"] = control_image.size[1]
            pipe_args["width"] = control_image.size[0]
        else:
            pipe_args["image"] = control_image
            pipe_args["controlnet_conditioning_scale"] = control_cond_scale
    if SDEdit:                                             # :274-276
        pipe_args["image"] = orig_img
        pipe_args["strength"] = SDEdit_strength
    output = pipe(**pipe_args)
    return output.images[0]


# ------------------------------------------------------------------------------------------
# planning pass: replay of the reference's host RNG (run_aug/run_aug.py:357-434)
# ------------------------------------------------------------------------------------------
def decorate_prompt(s: Settings, prompt, i, image_stem, source_image_path, image_classes_dict, ds_utils=None):
    """Car-part prefix, artistic / camera suffixes and sub-class injection, consuming the python / numpy RNG streams
    exactly like run_aug/run_aug.py:386-427 (per dataset: planes / cars look the class up by image stem, dtd / cub /
    compcars-parts by image path)."""
    if s.DATASET == "compcars-parts":                         # :387-390 the photographed part goes in front
        part = source_image_path.split("/")[-2]
        prompt = f"{ds_utils.get_basic_prompt(part=part)} {prompt}"
    if s.USE_ARTISTIC_PROMPTS and ((i % 2 == 0 and s.ARTISTIC_PROMPTS_PROB == 0.5) or
                                   (random.random() < s.ARTISTIC_PROMPTS_PROB and s.ARTISTIC_PROMPTS_PROB != 0.5)):
        prompt = f"{prompt}, {np.random.choice(ARTISTIC_PROMPTS)}"
    elif s.USE_CAMERA_VARIATIONS_PROMPTS and random.random() < s.CAMERA_VAIRATIONS_PROB:
        prompt = f"{prompt}, {np.random.choice(IMAGE_VARIATIONS_PROMPTS)} photo"
    if s.PROMPT_WITH_SUB_CLASS:
        if s.DATASET in ["planes", "planes_biased", "synthetic"]:
            prompt = prompt.replace("airplane", f"{image_classes_dict[image_stem]} airplane")
        elif s.DATASET == "cars":
            prompt = prompt.replace("car", f"{image_classes_dict[image_stem]} car")
        elif s.DATASET == "dtd":
            prompt = f"{prompt} with a {image_classes_dict[source_image_path]} texture"
        elif s.DATASET in ("compcars", "compcars-parts"):
            prompt = prompt.replace("car", f"{image_classes_dict[source_image_path]} car")
        elif s.DATASET == "cub":
            prompt = prompt.replace("bird", f"{image_classes_dict[source_image_path]} bird")
        else:
            raise NotImplementedError(s.DATASET)
    return prompt


def default_prompts_file(s: Settings):
    """The prompt source the reference's config block selects per dataset (run_aug/run_aug.py:589-666) for
    PROMPT_TYPE "gpt-meta_class" (100 GPT-written prompts per meta class, shipped as data next to this module) or
    "captions" (per-image BLIP captions: DTD only), and for the baseline prompt sources "ALIA" / "txt2sentence" /
    "txt2sentence-per_class" where the reference ships the file."""
    here = Path(__file__).parent / "prompts_engineering"
    if s.PROMPT_TYPE == "captions":
        if s.DATASET != "dtd":
            raise NotImplementedError(f"no caption file ships for {s.DATASET} (the reference has dtd_captions.json only)")
        return str(here / "captions" / "dtd_captions.json")
    if s.PROMPT_TYPE == "ALIA":
        # run_aug/run_aug.py:665-666 (the reference forgets the f-string prefix there; the files it means ship with it)
        f = here / "ALIA_prompts" / "gpt_output" / f"{'planes' if s.DATASET == 'synthetic' else s.DATASET}_prompts.txt"
        if not f.exists():
            raise NotImplementedError(f"no ALIA prompt file for {s.DATASET}")
        return str(f)
    if s.PROMPT_TYPE in ("txt2sentence", "txt2sentence-per_class"):
        # run_aug/run_aug.py:592-660 names LE_{200,30}_{dataset}_all_classes_{False,True}.json; the reference ships only the
        # cars txt2sentence file (under prompts_engineering/txt2sentance_prompts): anything else comes through PROMPTS_FILE
        ds = {"compcars-parts": "cars", "planes_biased": "planes"}.get(s.DATASET, s.DATASET)
        name = f"LE_30_{ds}_all_classes_True.json" if s.PROMPT_TYPE.endswith("per_class") else f"LE_200_{ds}_all_classes_False.json"
        f = here / "txt2sentance_prompts" / name
        if not f.exists():
            raise NotImplementedError(f"{s.PROMPT_TYPE}: {name} does not ship with the reference; pass Settings.PROMPTS_FILE "
                                      "(a {class: [prompts]} JSON written by prompts_engineering/txt2sentance_prompts.py)")
        return str(f)
    if s.PROMPT_TYPE != "gpt-meta_class":
        raise NotImplementedError(f"PROMPT_TYPE {s.PROMPT_TYPE}")
    name = {"planes": "planes", "planes_biased": "planes", "synthetic": "planes", "cars": "cars", "compcars-parts": "cars",
            "cub": "cub"}.get(s.DATASET)
    if name is None:
        raise NotImplementedError(f"no gpt-meta_class prompt file for {s.DATASET}")
    return str(here / "gpt_prompts" / f"{name}-100-gpt_v1.txt")


def _png_orientation(path, im=None):
    """EXIF orientation of a PNG without decoding it: the eXIf chunk may sit before or after the IDAT chunks, so the chunk
    headers are walked (8 bytes each, data skipped by seek) up to IEND.  1 when there is none.
    Pillow's `getexif()` -- what `ImageOps.exif_transpose` in `load_raw` consults -- also takes the orientation from a
    `Raw profile type exif` tEXt / zTXt chunk (ImageMagick-written PNGs) and from an XMP packet's `tiff:Orientation`
    (iTXt): a PNG that has no eXIf chunk but carries ANY text chunk is therefore handed to Pillow itself (`im`, or the
    file re-opened), so that the plan and the loader can never disagree; the fast path stays for PNGs with neither."""
    import struct
    text_chunk = False
    with open(path, "rb") as f:
        if f.read(8) != b"\x89PNG\r\n\x1a\n":
            return 1
        while True:
            head = f.read(8)
            if len(head) < 8:
                break
            n, kind = struct.unpack(">I4s", head)
            if kind == b"eXIf":
                ex = Image.Exif()
                ex.load(f.read(n))
                return ex.get(0x0112, 1)
            if kind in (b"tEXt", b"zTXt", b"iTXt"):
                text_chunk = True
            if kind == b"IEND":
                break
            f.seek(n + 4, 1)
    if not text_chunk:
        return 1
    if im is not None:
        return im.getexif().get(0x0112, 1)
    with Image.open(path) as im2:
        return im2.getexif().get(0x0112, 1)


def plan_work(s: Settings, original_images_paths, prompts, output_folder, image_classes_dict, image_size_fn=None,
              same_class_fn=None, ds_utils=None, captions=None, class_to_prompts=None):
    """Returns the work items in the reference's loop order.  Must be called right after
    utils.set_seed(SEED) (and dataset construction), like the reference's loop."""
    if image_size_fn is None:
        def image_size_fn(path):
            # the size AFTER the EXIF transpose load_source applies (diffusers.utils.load_image does the same): orientations
            # 5-8 swap the sides, and the plan's bucket / noise shape must be the loaded image's
            with Image.open(path) as im:
                w, h = im.size
                # PNG: Pillow's getexif() DECODES the picture to reach an eXIf chunk behind the pixel data (20 ms per
                # 1024 x 683 source: 69 s of a 3 334-image plan, tools/config3_rehearsal.py); walk the chunk headers instead
                orientation = _png_orientation(path, im) if im.format == "PNG" else im.getexif().get(0x0112, 1)
                if orientation in (5, 6, 7, 8):
                    w, h = h, w
            th, tw, _ = utils.resize_target_size(h, w, s.RESOLUTION)
            return th, tw
    if s.DEBUG:
        if s.SPECIFIC_FILE_STRs:
            original_images_paths = [p for p in original_images_paths if any(x in p for x in s.SPECIFIC_FILE_STRs)]
        else:
            original_images_paths = original_images_paths[:4]
    prompts = [p[:-1] if p and p[-1] == "." else p for p in (prompts or [])]        # :380 (idempotent)
    items = []
    noise_cursor = 0
    for index, source_image_path in enumerate(original_images_paths):
        image_stem = Path(source_image_path).stem
        th, tw = image_size_fn(source_image_path)
        if s.PROMPT_TYPE == "captions":                                      # :361-363 the image's own BLIP caption, N times
            cap = captions[source_image_path]["caption"][:MAX_PROMPT_LENGTH]
            prompts = [cap[:-1] if cap and cap[-1] == "." else cap] * s.NUM_PER_IMAGE
        elif s.PROMPT_TYPE == "txt2sentence-per_class":                        # :365-367 the prompts of the image's own class
            key = image_stem if s.DATASET in ("planes", "cars", "planes_biased", "synthetic") else source_image_path
            prompts = [q[:-1] if q and q[-1] == "." else q for q in class_to_prompts[image_classes_dict[key]]]
        sampled = np.random.choice(prompts, s.NUM_PER_IMAGE)                 # :382
        for i, prompt in enumerate(sampled):
            prompt = decorate_prompt(s, str(prompt), i, image_stem, source_image_path, image_classes_dict, ds_utils)
            output_path = Path(output_folder) / f"{image_stem[:MAX_FILENAME_LENGTH]}_prompt_{prompt.replace('/', '-')}_{i}.png"
            it = WorkItem(len(items), index, source_image_path, image_stem, i, prompt, str(output_path), th, tw)
            if output_path.exists():
                it.skip = True               # :430-432 -- skipped BEFORE the subject / noise draws
            else:
                if "blip_diffusion" in s.BASE_MODEL and s.STYLE_IMG_FROM_DIFF_IMG:
                    it.subject_path = random.choice(same_class_fn(source_image_path))     # :446 (python RNG stream)
                it.noise_offset = noise_cursor
                noise_cursor += 4 * (th // 8) * (tw // 8)
            items.append(it)
    return items


def shard_items(items, world):
    """Contiguous blocks balanced by sum(H*W) of the items that will actually run."""
    live = [it for it in items if not it.skip]
    total = sum(it.height * it.width for it in live)
    shards = [[] for _ in range(world)]
    acc, r = 0, 0
    for it in live:
        if r < world - 1 and acc >= (r + 1) * total / world:
            r += 1
        shards[r].append(it)
        acc += it.height * it.width
    return shards


_NOISE_SKIP_EXACT = {}


def _noise_skip_is_exact(dtype):
    """Can the CPU generator be advanced past n normal draws by drawing n bytes instead?  torch's CPU `normal_` fills a contiguous tensor
    of >= 16 elements with ONE 32-bit generator output per element (uniforms, then Box-Muller in place on blocks of 16), and `random_()` on
    a uint8 tensor also takes one output per element -- an implementation detail, so it is CHECKED once per process and dtype on a small
    case; if it ever stops holding, the replay below simply draws every item as before."""
    if dtype not in _NOISE_SKIP_EXACT:
        shape = (1, 4, 8, 10)                                   # 320 elements: a multiple of 16, not a power of two
        g = torch.Generator().manual_seed(1234)
        ref = [torch.randn(shape, generator=g, dtype=dtype) for _ in range(3)][2]
        g2 = torch.Generator().manual_seed(1234)
        try:
            torch.empty(2 * 320, dtype=torch.uint8).random_(generator=g2)
            _NOISE_SKIP_EXACT[dtype] = bool(torch.equal(torch.randn(shape, generator=g2, dtype=dtype), ref))
        except Exception:                                       # noqa: BLE001 -- any surprise means "do not skip"
            _NOISE_SKIP_EXACT[dtype] = False
    return _NOISE_SKIP_EXACT[dtype]


def noise_for_items(items_all, mine, seed, dtype, draws=1):
    """Replays the single sequential CPU noise stream (generator = torch.manual_seed(SEED),
    run_aug/run_aug.py:324) in work-item order and returns {order: [draws,4,h,w]} for `mine`.  draws = 2 for SDEdit:
    the img2img pipeline draws the posterior sample and then the scheduler noise for every item.
    Items of other ranks are not drawn but SKIPPED (round 6): the generator is advanced by their element count with byte draws, 7x
    faster than the normal transform (rank 7 of configs[3]: 10.5 s -> 1.5 s at start-up), bit-identical where `_noise_skip_is_exact`
    holds and every skipped tensor has a multiple of 16 elements (always: sides are multiples of 64)."""
    want = {it.order for it in mine}
    last = max(want) if want else -1
    g = torch.manual_seed(seed)
    out = {}
    can_skip = _noise_skip_is_exact(dtype) and os.environ.get("SASPA_NOISE_SKIP", "1") != "0"
    pending = 0
    scratch = None

    def flush(count):
        nonlocal scratch
        chunk = 1 << 24
        if scratch is None:
            scratch = torch.empty(min(count, chunk), dtype=torch.uint8)
        if scratch.numel() < min(count, chunk):
            scratch = torch.empty(min(count, chunk), dtype=torch.uint8)
        while count > 0:
            n = min(count, chunk)
            scratch[:n].random_(generator=g)
            count -= n

    for it in items_all:
        if it.skip:
            continue
        if it.order > last:
            break
        numel = 4 * (it.height // 8) * (it.width // 8)
        if it.order not in want and can_skip and numel % 16 == 0:
            pending += draws * numel
            continue
        if pending:
            flush(pending)
            pending = 0
        n = torch.cat([torch.randn((1, 4, it.height // 8, it.width // 8), generator=g, dtype=dtype) for _ in range(draws)])
        if it.order in want:
            out[it.order] = n
    return out


def make_batches(mine, batch_size):
    buckets = {}
    for it in mine:
        buckets.setdefault((it.height, it.width), []).append(it)
    batches = []
    for key in sorted(buckets):
        b = buckets[key]
        batches += [b[j:j + batch_size] for j in range(0, len(b), batch_size)]
    return batches


# ------------------------------------------------------------------------------------------
# generation of one batch of work items on the GPU
# ------------------------------------------------------------------------------------------
def make_hed_detector(s: Settings, device):
    """run_aug/run_aug.py:311-312: `HEDdetector.from_pretrained('lllyasviel/ControlNet')` -- from WEIGHTS_DIR when given,
    architecture-exact synthetic weights otherwise (no checkpoint offline)."""
    from . import weights as W
    from .config import HED
    from .hed import HEDdetector
    if s.WEIGHTS_DIR:
        return HEDdetector.from_pretrained(os.path.join(s.WEIGHTS_DIR, "lllyasviel/ControlNet"), device=device)
    logging.info("no WEIGHTS_DIR given: HED annotator with architecture-exact SYNTHETIC weights")
    return HEDdetector(W.synth_state_dict("hed", HED, 7), HED, device)


def hip_batch_generator(pipe, s: Settings):
    """Returns fn(batch_items, noises[list of [1,4,h,w]], source_u8 [B,H,W,3]) -> (images u8
    [B,H,W,3] numpy, control u8 [B,H,W,3] numpy), running Canny + sampling on the device."""
    from . import ops
    tok = pipe.tokenizer
    neg_ids = tok(NEGATIVE_PROMPT)

    blip = "blip_diffusion" in s.BASE_MODEL
    hed = make_hed_detector(s, pipe.device) if s.CONTROLNET == "hed" else None

    side = torch.cuda.Stream(device=pipe.device)

    def to_device(raw, height, width):
        """One decoded source image (host u8 [h,w,3], any size) -> device u8 [height,width,3]: `resize_image`
        (all_utils/utils.py:58-79) on the device, stream-ordered -- nothing here waits for the GPU."""
        from . import imageproc
        d = ops.h2d(torch.from_numpy(np.ascontiguousarray(raw)), pipe.device)
        if tuple(d.shape[:2]) == (height, width):
            return d
        _, _, k = utils.resize_target_size(d.shape[0], d.shape[1], s.RESOLUTION)
        return imageproc.cv_resize_u8(d[None].contiguous(), height, width, "lanczos4" if k > 1 else "area")[0]

    def enqueue(batch, noises, sources, subjects=None, category=None):
        """Everything of one batch up to the device-resident u8 images, WITHOUT waiting for the GPU.  `sources` /
        `subjects`: decoded images as loaded (lists; a stacked array when they already have the planned size)."""
        src = torch.stack([to_device(raw, it.height, it.width) for raw, it in zip(sources, batch)])
        if not s.CONTROLNET:                                                   # run_aug/run_aug.py:435: no control image at all
            ctrl = None
        elif hed is not None:                                                  # run_aug/run_aug.py:438-439
            ctrl = hed.detect_batch(src)
        else:
            ctrl = ops.canny(src, s.LOW_THRESHOLD_CANNY, s.HIGH_THRESHOLD_CANNY)   # always from the ORIGINAL image (:437)
        lat = torch.cat(noises)
        subs = None
        if blip:
            # subject tokens from the same-class image (or the image itself), amplified "a {category} {prompt}" prompt
            # tokenised to 77 - 16 tokens, no conditioning scale (run_aug/run_aug.py:243-250, 262-265, 444-456)
            if subjects is not None:
                subs = []
                for raw in subjects:
                    th, tw, _ = utils.resize_target_size(raw.shape[0], raw.shape[1], s.RESOLUTION)
                    subs.append(to_device(raw, th, tw))
            refs = subs if subs is not None else list(src)
            q = pipe.get_query_embeddings(refs, [category] * len(batch))
            ids = np.concatenate([tok(pipe.build_prompt(it.prompt, category), max_len=pipe.prompt_token_count()) for it in batch])
            out = pipe.generate_batch(ids, neg_ids, ctrl, lat, s.NUM_INFERENCE_STEPS, s.GUIDANCE_SCALE, 1.0, query_embeds=q)
        elif s.SDEDIT:
            # SDEdit (run_aug/run_aug.py:252-260, :274-276): img2img from the source image itself, two draws per item
            ids = np.concatenate([tok(it.prompt) for it in batch])
            if ctrl is None:                               # Real-Guidance: StableDiffusionImg2ImgPipeline, UNet only
                out = pipe.generate_batch_img2img(ids, neg_ids, src, lat[0::2], lat[1::2], s.NUM_INFERENCE_STEPS,
                                                  s.SDEDIT_STRENGTH, s.GUIDANCE_SCALE)
            else:
                out = pipe.generate_batch_img2img(ids, neg_ids, src, ctrl, lat[0::2], lat[1::2], s.NUM_INFERENCE_STEPS,
                                                  s.SDEDIT_STRENGTH, s.GUIDANCE_SCALE, s.CONTROLNET_CONDITIONING_SCALE)
        else:
            ids = np.concatenate([tok(it.prompt) for it in batch])
            out = pipe.generate_batch(ids, neg_ids, ctrl, lat, s.NUM_INFERENCE_STEPS, s.GUIDANCE_SCALE,
                                      s.CONTROLNET_CONDITIONING_SCALE)
        ev = torch.cuda.Event()
        ev.record()
        return out, ctrl, ev, src, subs

    def finish(handle):
        """Device -> host copies of a batch enqueued earlier, on a side stream that waits for THAT batch only: the next
        batch's launch sequence may already be queued behind it on the main stream and keeps the GPU busy meanwhile.
        -> (images, controls, resized sources, resized subjects | None)."""
        out, ctrl, ev, src, subs = handle
        side.wait_event(ev)
        with torch.cuda.stream(side):
            for t in (out, src) + ((ctrl,) if ctrl is not None else ()) + tuple(subs or ()):
                t.record_stream(side)
            if os.environ.get("SASPA_HOST_BLOCKING", "1") == "0":
                o, sr = out.cpu(), src.cpu()                    # (a synchronous copy spins on the calling thread until the batch is done)
                c = ctrl.cpu() if ctrl is not None else None
                sb = [t.cpu() for t in subs] if subs is not None else None
            else:
                # asynchronous copies into pinned buffers, then ONE sleeping wait (event query + 1 ms naps, ops.sleep_wait): the thread
                # gives its core back for the rest of the batch instead of spinning in hipMemcpy / hipEventSynchronize (round 6: host
                # budget of 8 ranks on a 16-core quota)
                def d2h(t):
                    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    h.copy_(t, non_blocking=True)
                    return h
                o, sr = d2h(out), d2h(src)
                c = d2h(ctrl) if ctrl is not None else None
                sb = [d2h(t) for t in subs] if subs is not None else None
                done = torch.cuda.Event()
                done.record()
                ops.sleep_wait(done, 0.002)
        return o.numpy(), (c.numpy() if c is not None else None), sr.numpy(), ([t.numpy() for t in sb] if sb is not None else None)

    def run(batch, noises, sources, subjects=None, category=None):
        return finish(enqueue(batch, noises, sources, subjects, category))

    run.enqueue, run.finish = enqueue, finish
    return run


def load_raw(path):
    """diffusers.utils.load_image (run_aug/run_aug.py:372): open, EXIF-transpose, RGB -> u8 [h,w,3]."""
    from PIL import ImageOps
    return np.array(ImageOps.exif_transpose(Image.open(path)).convert("RGB"))


def load_source(path, resolution):
    """diffusers.utils.load_image + utils.resize_image (run_aug/run_aug.py:372-374) for ONE image (the generation loop
    resizes on the device inside the batch instead: hip_batch_generator.to_device)."""
    return utils.resize_image(load_raw(path), resolution)


def host_threads(local_world=1):
    """Intra-op threads a generation rank gives torch's CPU pool: min(4, CPU quota / ranks on this node), SASPA_HOST_THREADS
    overrides.  torch sizes the pool to the machine (128 threads on the 256-logical-CPU MI355X hosts, whatever the cgroup quota);
    the loop's small CPU tensor ops (noise slices, pinned staging copies, token arrays) woke all of them and each spun in OpenMP's
    wait loop afterwards: 128 x 2.7 s = 0.21 CPU-s per image, a third of the main process' CPU time
    (profiles/r6_config3_full_shard.json).  Nothing on the generation path is CPU-compute-bound."""
    env = os.environ.get("SASPA_HOST_THREADS")
    if env:
        return max(1, int(env))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(4, n // max(1, int(local_world))))


def main(s: Settings, ds_utils=None, batch_generator=None, dist=None, pipe=None, filter_models=None):
    """`_main` with torch's CPU pool held at `host_threads()` for the duration of the run (restored afterwards)."""
    world = dist.get_world_size() if dist is not None else 1
    before = torch.get_num_threads()
    torch.set_num_threads(host_threads(int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    try:
        return _main(s, ds_utils=ds_utils, batch_generator=batch_generator, dist=dist, pipe=pipe, filter_models=filter_models)
    finally:
        torch.set_num_threads(before)


def _main(s: Settings, ds_utils=None, batch_generator=None, dist=None, pipe=None, filter_models=None):
    """The generation loop.  `batch_generator` is injectable for host-logic tests; the default
    builds the HIP pipeline (fails loudly without an MI355X).  `filter_models` = (SemanticFilter | None,
    ConfidenceFilter | None) to reuse built filter models; None builds them on s.DEVICE when a filter flag is set."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    png = _PngWriters(4)
    utils.set_seed(s.SEED)
    if ds_utils is None:
        ds_utils = dataset_utils.DS_UTILS_DICT[s.DATASET](**s.DATASET_KWARGS)
    if s.DATASET == "dtd" and s.PROMPT_TYPE != "captions":        # run_aug/run_aug.py:611-616: DTD runs on captions only
        logging.info("DTD is generated from per-image captions: PROMPT_TYPE -> captions")
        s.PROMPT_TYPE = "captions"
    prompts_file = s.PROMPTS_FILE or default_prompts_file(s)
    output_folder = output_folder_for(s, ds_utils.root_path)
    if rank == 0:
        Path(output_folder).mkdir(parents=True, exist_ok=True)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        utils.init_logging(str(Path(output_folder).parent))
    image_classes_dict = ds_utils.get_image_stem_to_class_str_dict()     # path-keyed for dtd / cub / compcars-parts (:700)
    captions, class_to_prompts = None, None
    if s.PROMPT_TYPE == "captions":
        import json as _json
        with open(prompts_file, "r") as f:
            captions = _json.load(f)
        prompts = None
        logging.info(f"Read {len(captions)} captions from {prompts_file}")
    elif s.PROMPT_TYPE == "txt2sentence-per_class":
        class_to_prompts = read_prompts_from_json(prompts_file, s.DATASET, per_class=True)
        prompts = None
        logging.info(f"Read prompts json with {len(class_to_prompts)} classes from {prompts_file}")
    elif s.PROMPT_TYPE == "txt2sentence":
        prompts = read_prompts_from_json(prompts_file, s.DATASET, per_class=False)
        logging.info(f"Read {len(prompts)} prompts from {prompts_file}")
    else:                                           # gpt-meta_class, ALIA: one prompt per line
        prompts = read_prompts(prompts_file)
        logging.info(f"Read {len(prompts)} prompts from {prompts_file}")
    aug_json_path = utils.get_aug_json_path(output_folder, semantic_filtering=s.SEMANTIC_FILTERING,
                                            model_confidence_based_filtering=s.MODEL_CONFIDENCE_BASED_FILTERING)
    logging.info(f"Augmented json path will be at: \n{aug_json_path}")
    if filter_models is None and (s.SEMANTIC_FILTERING or s.MODEL_CONFIDENCE_BASED_FILTERING):
        # fail BEFORE hours of generation, not after: a filter flag with no checkpoint behind it is an error unless
        # synthetic filter weights were asked for explicitly (SASPA_SYNTHETIC_FILTERS=1)
        from . import filters as _filters
        _filters.filter_checkpoints(ds_utils, s.WEIGHTS_DIR, bool(s.SEMANTIC_FILTERING), bool(s.MODEL_CONFIDENCE_BASED_FILTERING))

    blip = "blip_diffusion" in s.BASE_MODEL
    items = plan_work(s, ds_utils.original_images_paths, prompts, output_folder, image_classes_dict,
                      same_class_fn=ds_utils.get_image_path_with_same_class if blip else None, ds_utils=ds_utils,
                      captions=captions, class_to_prompts=class_to_prompts)
    mine = shard_items(items, world)[rank]
    logging.info(f"rank {rank}/{world}: {len(mine)} of {len(items)} work items ({sum(i.skip for i in items)} already exist)")

    if batch_generator is None:
        if pipe is None:
            pipe = init_pipeline(s.BASE_MODEL, s.CONTROLNET, s.SDEDIT, weights_dir=s.WEIGHTS_DIR)
            pipe = pipe.to(s.DEVICE, torch.float32 if s.PRECISION == "fp32" else torch.float16)
        batch_generator = hip_batch_generator(pipe, s)
        noise_dtype = pipe.noise_dtype
    else:
        noise_dtype = torch.float32 if s.PRECISION == "fp32" else torch.float16
    noises = noise_for_items(items, mine, s.SEED, noise_dtype, draws=2 if s.SDEDIT else 1)   # generator = torch.manual_seed(SEED) (:324)

    first_variant = {}
    for it in items:
        first_variant.setdefault(it.index, it.order)
    num_errors = 0

    def load_batch(batch):
        """Decode the batch's source (and subject) images: host work, prefetched one batch ahead on its own thread so the
        GPU does not wait for JPEG decoding between launch sequences (the resize runs on the device, in `enqueue`)."""
        srcs = [load_raw(it.source_path) for it in batch]
        subs = [load_raw(it.subject_path) if it.subject_path else srcs[k] for k, it in enumerate(batch)] if blip else None
        if all(r.shape[:2] == (it.height, it.width) for r, it in zip(srcs, batch)):
            srcs = np.stack(srcs)            # already at the planned size: what injected (host-only) generators consume
        return srcs, subs

    loader = ThreadPoolExecutor(max_workers=1)
    batches = make_batches(mine, s.BATCH_SIZE)
    pending = loader.submit(load_batch, batches[0]) if batches else None
    # software pipeline over batches: batch i+1 is enqueued on the GPU BEFORE batch i's images are copied back and handed
    # to the PNG pool, so the GPU never waits for host-side post-processing (injected test generators run synchronously)
    pipelined = hasattr(batch_generator, "enqueue") and hasattr(batch_generator, "finish")
    inflight = None

    def failed(batch, e):
        nonlocal num_errors
        logging.exception(e)                 # the reference treats RuntimeError as OOM (:493-500); isolate per batch
        num_errors += 1
        for it in batch:
            it.status = -1

    def drain(entry):
        batch, handle, sources, subjects = entry
        try:
            images, controls, resized, resized_subjects = batch_generator.finish(handle)
        except RuntimeError as e:
            failed(batch, e)
            return False
        emit(batch, images, controls, resized, resized_subjects if subjects is not None else None)
        batch_log.append((batch[0].height, batch[0].width, len(batch), _time.time()))
        return True

    def emit(batch, images, controls, sources, subjects):
        for k, it in enumerate(batch):
            stem40 = it.image_stem[:MAX_FILENAME_LENGTH]
            if first_variant[it.index] == it.order:
                png.submit(sources[k], os.path.join(output_folder, f"{stem40}_source.png"))
                if it.index < 10 and controls is not None:       # :441-442 sits inside `if CONTROLNET:`
                    png.submit(controls[k], f"{output_folder}/{stem40}_control.png")
            if subjects is not None and it.subject_path:     # :453-454 "_subject_{i}.png" (excluded from the JSON by name)
                png.submit(subjects[k], os.path.join(output_folder, f"{stem40}_subject_{it.i}.png"))
            png.submit(images[k], it.output_path)
            it.status = 1

    import time as _time
    prof = os.environ.get("SASPA_PROFILE_LOOP") == "1"          # per-batch host timings in the log (diagnostics)
    batch_log = []                                               # per drained batch: (height, width, items, time drained)
    if s.MAX_BATCHES > 0:
        batches = batches[:s.MAX_BATCHES]
    for bi, batch in enumerate(batches):
        try:
            t0 = _time.time()
            fut, pending = pending, (loader.submit(load_batch, batches[bi + 1]) if bi + 1 < len(batches) else None)
            sources, subjects = fut.result()
            t1 = _time.time()
            args = (batch, [noises[it.order] for it in batch], sources) + ((subjects, ds_utils.meta_class) if blip else ())
            if pipelined:
                handle = batch_generator.enqueue(*args)
            else:
                images, controls = batch_generator(*args)
            if prof:
                print(f"[loop] batch {bi}: wait for sources {t1 - t0:.3f} s, enqueue {_time.time() - t1:.3f} s", flush=True)
        except KeyboardInterrupt:
            raise
        except RuntimeError as e:
            failed(batch, e)
            if num_errors > 20:
                logging.info("Too many errors, stopping generation on this rank")
                break
            continue
        if pipelined:
            if inflight is not None:
                t2 = _time.time()
                ok = drain(inflight)
                if prof:
                    print(f"[loop] batch {bi}: drain of the previous batch {_time.time() - t2:.3f} s", flush=True)
                if not ok and num_errors > 20:       # asynchronous failures surface here: same abort rule (:497-500)
                    logging.info("Too many errors, stopping generation on this rank")
                    # the batch enqueued just now is drained like any other: written and marked when it succeeded,
                    # counted as failed when it did not (it used to be finished and then dropped, status 0)
                    drain((batch, handle, sources, subjects))
                    inflight = None
                    break
            inflight = (batch, handle, sources, subjects)
        else:
            emit(batch, images, controls, sources, subjects)
    if inflight is not None:
        drain(inflight)
    png.close()
    loader.shutdown()

    # ---- the one collective: per-item status vector -> rank 0 ----
    status = torch.zeros(len(items), dtype=torch.int32)
    for it in mine:
        status[it.order] = it.status
    if dist is not None:
        dev = torch.device(s.DEVICE) if dist.get_backend() == "nccl" else torch.device("cpu")
        st = status.to(dev)
        gathered = [torch.empty_like(st) for _ in range(world)] if rank == 0 else None
        dist.gather(st, gathered, dst=0)
        if rank == 0:
            status = torch.stack([g.cpu() for g in gathered]).sum(0).to(torch.int32)   # shards are disjoint
    json_path = None
    if rank == 0:
        logging.info(f"Done Generating: {(status == 1).sum().item()} generated, {(status == -1).sum().item()} failed, "
                     f"{sum(i.skip for i in items)} skipped (already existed)")
        n_files = len(list(Path(output_folder).glob("*.*")))
        # the filter stage runs on this rank's GPU whether the models are built here or handed in
        fdev = torch.device(s.DEVICE) if (s.SEMANTIC_FILTERING or s.MODEL_CONFIDENCE_BASED_FILTERING) else None
        json_path = utils.create_json_of_image_name_to_augmented_images_paths(
            ds_utils, output_folder, semantic_filtering=s.SEMANTIC_FILTERING,
            model_confidence_based_filtering=s.MODEL_CONFIDENCE_BASED_FILTERING, init_log=False,
            original_images_paths=ds_utils.original_images_paths, min_files=min(10, max(1, n_files)),
            filter_models=filter_models, weights_dir=s.WEIGHTS_DIR, device=fdev)
    if dist is not None:
        dist.barrier()
    return dict(items=items, status=status, json_path=json_path, output_folder=output_folder, mine=mine, n_batches=len(batches),
                png_submitted=png.submitted, png_max_queue=png.max_depth, batch_log=batch_log)

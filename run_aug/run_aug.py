#!/usr/bin/env python3
"""Entry point with the reference's interface: edit the constants below (same names as
the reference's run_aug/run_aug.py:513-556) and run

    python run_aug/run_aug.py                                   # one MI355X
    SASPA_GPUS=8 python run_aug/run_aug.py                      # 8 MI355X: this script starts the 8 ranks itself
    SASPA_SUPERVISE=1 python run_aug/run_aug.py                 # one MI355X under the supervising launcher (see below)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run_aug/run_aug.py   # same, via torchrun

Started through this script's own launcher (SASPA_GPUS > 1, or SASPA_SUPERVISE=1 for a single GPU) the run is supervised:
if a rank dies on a signal -- the HIP runtime's graph-replay crash of long sessions, DESIGN.md section 7 -- all ranks are
started again ONCE as fresh processes with SASPA_FORK=0 (single-branch step graphs) and continue from the files that
exist; the exit code is non-zero if that fails too.  SASPA_FORK=0 by hand rules the two-branch capture out from the start.

Environment overlays (optional): SASPA_DATASET, SASPA_WEIGHTS_DIR, SASPA_PROMPTS_FILE,
SASPA_NUM_INFERENCE_STEPS, SASPA_NUM_PER_IMAGE, SASPA_PRECISION, SASPA_BASE_MODEL (sd_v1.5 | blip_diffusion | sd_xl-turbo)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import saspa_aug_amd  # noqa: E402,F401

if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and \
        (int(os.environ.get("SASPA_GPUS", "1")) > 1 or os.environ.get("SASPA_SUPERVISE", "0") == "1"):
    # become the launcher BEFORE torch is imported: the parent never touches the GPU (saspa_aug_amd/launcher.py)
    from saspa_aug_amd.launcher import launch_supervised  # noqa: E402
    sys.exit(launch_supervised(int(os.environ.get("SASPA_GPUS", "1")), __file__, sys.argv[1:]))

from saspa_aug_amd import run_aug as R  # noqa: E402

if __name__ == "__main__":
    DEBUG = 0
    SPECIFIC_FILE_STRs = None
    # ---------------------------- generation params ----------------------------
    DEVICE = "cuda:0"
    version = "v1"
    DATASET = os.environ.get("SASPA_DATASET", "planes")
    BASE_MODEL = os.environ.get("SASPA_BASE_MODEL") or ("sd_v1.5" if DATASET in ("planes", "synthetic") else "blip_diffusion")
    CONTROLNET = "canny"
    SDEDIT = 0
    NUM_PER_IMAGE = int(os.environ.get("SASPA_NUM_PER_IMAGE", 2))
    SEED = 1
    PROMPT_TYPE = "gpt-meta_class"
    PROMPT_WITH_SUB_CLASS = True
    USE_ARTISTIC_PROMPTS = True if BASE_MODEL == "sd_v1.5" else False
    ARTISTIC_PROMPTS_PROB = 0.5
    USE_CAMERA_VARIATIONS_PROMPTS = False
    CAMERA_VAIRATIONS_PROB = 0.5
    RESOLUTION = 512
    GUIDANCE_SCALE = 7.5
    NUM_INFERENCE_STEPS = int(os.environ.get("SASPA_NUM_INFERENCE_STEPS", 30))
    LOW_THRESHOLD_CANNY = 120
    HIGH_THRESHOLD_CANNY = 200
    CONTROLNET_CONDITIONING_SCALE = 0.75
    # ---------------------------- json creation params ----------------------------
    SEMANTIC_FILTERING = 1
    MODEL_CONFIDENCE_BASED_FILTERING = 1
    # ---------------------------- this build ----------------------------
    BATCH_SIZE = 8

    if "cars" in DATASET.lower():
        NUM_INFERENCE_STEPS = 50
    if DATASET.lower() == "cub":                      # reference :564-565
        BASE_MODEL = "sd_xl-turbo"
    if BASE_MODEL == "sd_xl-turbo":                   # reference :567-571
        print("Using sd_xl-turbo, setting some params accordingly")
        GUIDANCE_SCALE = 0
        NUM_INFERENCE_STEPS = 2
        R.NEGATIVE_PROMPT = None

    s = R.Settings(DEBUG=DEBUG, SPECIFIC_FILE_STRs=SPECIFIC_FILE_STRs, DEVICE=DEVICE, version=version, DATASET=DATASET,
                   BASE_MODEL=BASE_MODEL, CONTROLNET=CONTROLNET, SDEDIT=SDEDIT, NUM_PER_IMAGE=NUM_PER_IMAGE, SEED=SEED,
                   PROMPT_TYPE=PROMPT_TYPE, PROMPT_WITH_SUB_CLASS=PROMPT_WITH_SUB_CLASS,
                   USE_ARTISTIC_PROMPTS=USE_ARTISTIC_PROMPTS, ARTISTIC_PROMPTS_PROB=ARTISTIC_PROMPTS_PROB,
                   USE_CAMERA_VARIATIONS_PROMPTS=USE_CAMERA_VARIATIONS_PROMPTS, CAMERA_VAIRATIONS_PROB=CAMERA_VAIRATIONS_PROB,
                   RESOLUTION=RESOLUTION, GUIDANCE_SCALE=GUIDANCE_SCALE, NUM_INFERENCE_STEPS=NUM_INFERENCE_STEPS,
                   LOW_THRESHOLD_CANNY=LOW_THRESHOLD_CANNY, HIGH_THRESHOLD_CANNY=HIGH_THRESHOLD_CANNY,
                   CONTROLNET_CONDITIONING_SCALE=CONTROLNET_CONDITIONING_SCALE, SEMANTIC_FILTERING=SEMANTIC_FILTERING,
                   MODEL_CONFIDENCE_BASED_FILTERING=MODEL_CONFIDENCE_BASED_FILTERING, BATCH_SIZE=BATCH_SIZE,
                   PRECISION=os.environ.get("SASPA_PRECISION", "bf16"), WEIGHTS_DIR=os.environ.get("SASPA_WEIGHTS_DIR"),
                   PROMPTS_FILE=os.environ.get("SASPA_PROMPTS_FILE"))
    assert s.DATASET in R.dataset_utils.DATASETS_SUPPORTED
    assert s.BASE_MODEL in R.BASE_MODEL_DICT.keys()
    assert s.NUM_PER_IMAGE > 0

    dist = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("SASPA_FORCE_DIST", "0") == "1":
        # one process per GPU, RCCL over xGMI (SASPA_FORCE_DIST=1: the same leg at world size 1 -- a one-GPU rehearsal)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        s.DEVICE = f"cuda:{local_rank}"
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    result = R.main(s, dist=dist)
    if dist is not None:
        dist.destroy_process_group()
    if result["json_path"]:
        print(result["json_path"])

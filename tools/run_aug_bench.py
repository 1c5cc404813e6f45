"""End-to-end rate of the generation loop (saspa_aug_amd.run_aug.main) on a synthetic FGVC-Aircraft-shaped dataset:
everything the reference's loop does per image -- PNG decode + resize, host -> device copies, Canny, sampling, the safety
checker, device -> host copies, PNG encode of outputs / sources / controls, the status gather and the JSON -- at the BASELINE
operating point (512x512 and the 512x704 size most Aircraft images resize to, 50 DDIM steps, 4 variants, batch 8).
usage: python tools/run_aug_bench.py [n_images] [steps]"""
import json, os, shutil, sys, tempfile, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import run_aug as R
n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
tmp = tempfile.mkdtemp(prefix="saspa_e2e_")
prompts = os.path.join(tmp, "prompts.txt")
open(prompts, "w").write("".join(f"an airplane flying over landscape number {k}.\n" for k in range(20)))
pipe = R.init_pipeline("sd_v1.5", "canny", 0).to("cuda:0", torch.float16)
for sizes in (((512, 512),), ((512, 704),)):
    root = os.path.join(tmp, f"ds_{sizes[0][1]}", "data")
    s = R.Settings(DATASET="synthetic", BASE_MODEL="sd_v1.5", RESOLUTION=512, NUM_INFERENCE_STEPS=steps, NUM_PER_IMAGE=4, SEED=1,
                   SEMANTIC_FILTERING=0, MODEL_CONFIDENCE_BASED_FILTERING=0, PROMPTS_FILE=prompts, BATCH_SIZE=8,
                   DATASET_KWARGS=dict(root_path=root, n_images=n_images, sizes=sizes))
    if sizes[0][1] == 512:          # warm the kernels / allocator on a throw-away run of one batch
        sw = R.Settings(**{**s.__dict__, "NUM_PER_IMAGE": 1, "NUM_INFERENCE_STEPS": 2,
                           "DATASET_KWARGS": dict(root_path=os.path.join(tmp, "warm", "data"), n_images=8, sizes=sizes)})
        R.main(sw, pipe=pipe)
    from saspa_aug_amd.dataset_utils import SyntheticUtils
    SyntheticUtils(root_path=root, n_images=n_images, sizes=sizes, print_func=lambda *a: None)              # datasets exist before
    SyntheticUtils(root_path=root + "_x3", n_images=3 * n_images, sizes=sizes, print_func=lambda *a: None)  # the clock starts
    torch.cuda.synchronize(); t0 = time.time()
    res = R.main(s, pipe=pipe)
    torch.cuda.synchronize(); dt = time.time() - t0
    n = int((res["status"] == 1).sum())
    # a second, 3x longer run on a fresh dataset: the slope between the two is the steady-state rate (planning, the first
    # load, the final PNG flush and the JSON are per-run constants)
    s3 = R.Settings(**{**s.__dict__, "DATASET_KWARGS": dict(root_path=root + "_x3", n_images=3 * n_images, sizes=sizes)})
    torch.cuda.synchronize(); t0 = time.time()
    res3 = R.main(s3, pipe=pipe)
    torch.cuda.synchronize(); dt3 = time.time() - t0
    n3 = int((res3["status"] == 1).sum())
    print(json.dumps({"workload": f"run_aug.main end to end, synthetic dataset {n_images} / {3 * n_images} images x 4 variants at "
                                  f"{sizes[0][0]}x{sizes[0][1]}, {steps} DDIM steps, batch 8, PNG I/O + safety checker + JSON included",
                      "images": [n, n3], "seconds": [round(dt, 2), round(dt3, 2)], "images_per_s": [round(n / dt, 3), round(n3 / dt3, 3)],
                      "steady_state_images_per_s": round((n3 - n) / (dt3 - dt), 3), "dtype": "bf16", "data": "synthetic"}), flush=True)
shutil.rmtree(tmp, ignore_errors=True)

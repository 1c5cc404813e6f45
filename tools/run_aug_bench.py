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
from saspa_aug_amd.dataset_utils import SyntheticUtils
MULTS = (1, 2, 3)
for sizes in (((512, 512),), ((512, 704),)):
    root = os.path.join(tmp, f"ds_{sizes[0][1]}", "data")
    s = R.Settings(DATASET="synthetic", BASE_MODEL="sd_v1.5", RESOLUTION=512, NUM_INFERENCE_STEPS=steps, NUM_PER_IMAGE=4, SEED=1,
                   SEMANTIC_FILTERING=0, MODEL_CONFIDENCE_BASED_FILTERING=0, PROMPTS_FILE=prompts, BATCH_SIZE=8,
                   DATASET_KWARGS=dict(root_path=root, n_images=n_images, sizes=sizes))
    # warm THIS shape's step graph (its key holds the sizes AND the step count), the kernels and the allocator on a throw-away
    # run of one batch with the SAME step count -- round 3 warmed 512x512 with 2 steps only, so the first timed run of each
    # size paid the 50-step graph capture and the slope between it and the second run came out above the GPU-only rate
    sw = R.Settings(**{**s.__dict__, "NUM_PER_IMAGE": 1,
                       "DATASET_KWARGS": dict(root_path=os.path.join(tmp, f"warm_{sizes[0][1]}", "data"), n_images=8, sizes=sizes)})
    R.main(sw, pipe=pipe)
    for m in MULTS:                                                      # datasets exist before the clock starts
        SyntheticUtils(root_path=f"{root}_x{m}", n_images=m * n_images, sizes=sizes, print_func=lambda *a: None)
    ns, dts = [], []
    for m in MULTS:                                                      # three run lengths on fresh datasets
        sm = R.Settings(**{**s.__dict__, "DATASET_KWARGS": dict(root_path=f"{root}_x{m}", n_images=m * n_images, sizes=sizes)})
        torch.cuda.synchronize(); t0 = time.time()
        res = R.main(sm, pipe=pipe)
        torch.cuda.synchronize(); dts.append(time.time() - t0)
        ns.append(int((res["status"] == 1).sum()))
    # least-squares slope of images over seconds through the three points = the steady-state rate (planning, the first load,
    # the final PNG flush and the JSON are per-run constants); it cannot exceed the GPU-only rate of bench.py on the same box
    mt, mn = sum(dts) / len(dts), sum(ns) / len(ns)
    slope = sum((t - mt) * (n - mn) for t, n in zip(dts, ns)) / sum((t - mt) ** 2 for t in dts)
    print(json.dumps({"workload": f"run_aug.main end to end, synthetic datasets of {[m * n_images for m in MULTS]} images x 4 variants at "
                                  f"{sizes[0][0]}x{sizes[0][1]}, {steps} DDIM steps, batch 8, PNG I/O + safety checker + JSON included; "
                                  "this shape's step graph warmed by a one-batch run first",
                      "images": ns, "seconds": [round(t, 2) for t in dts], "images_per_s": [round(n / t, 3) for n, t in zip(ns, dts)],
                      "steady_state_images_per_s": round(slope, 3), "steady_state_method": "least-squares slope over the three run lengths",
                      "dtype": "bf16", "data": "synthetic"}), flush=True)
shutil.rmtree(tmp, ignore_errors=True)

import os, time, torch, numpy as np, sys
sys.path.insert(0,'.')
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print("OMP", os.environ.get("OMP_NUM_THREADS"), "MKL", os.environ.get("MKL_NUM_THREADS"))
import saspa_aug_amd
from saspa_aug_amd import config as CFG, weights as W
from oracle import pipeline as OP
from oracle.canny import generate_canny_array
from saspa_aug_amd.synthetic import synthetic_image
def run(cfgs, fam, hh, steps):
    ids = torch.from_numpy(np.random.RandomState(1).randint(0, cfgs["text"]["vocab"] - 2, (1, 77)))
    neg = torch.from_numpy(np.random.RandomState(2).randint(0, cfgs["text"]["vocab"] - 2, (1, 77)))
    ctrl = generate_canny_array(synthetic_image(hh, hh, 10), 120, 200)
    lat = torch.randn((1,4,hh//8,hh//8))
    t=time.time(); OP.sd_controlnet_pipeline(fam, cfgs, ids, neg, ctrl, lat, steps, return_latents=True); return time.time()-t
cfgs = CFG.tiny(); fam = W.synth_family(cfgs, seed=3)
full = {k: v for k, v in CFG.SD15.items() if k != "safety"}; t=time.time(); ffam = W.synth_family(full, seed=0); print("synth full", time.time()-t)
for n in (torch.get_num_threads(), 64, 32, 16, 8):
    torch.set_num_threads(n)
    print("threads", n, "tiny 10 steps", round(run(cfgs, fam, 64, 10), 2), "full 256x256 1 step", round(run(full, ffam, 256, 1), 2), flush=True)

"""How many step graphs can one process capture before hipGraphLaunch dies?  (Round 5: a full `-m gpu` session ended in a
segmentation fault inside hipGraphLaunch at whichever tiny fp32 pipeline test came after ~N captures.)  Builds a tiny
SD + ControlNet pipeline over and over, each time capturing its step graph and replaying it for 10 steps, printing the count,
the process RSS and the device memory in use.  usage: python tools/graph_leak_probe.py [n=300] [dtype=fp32|bf16] [keep=0|1]"""
import gc, os, resource, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401,E402
from saspa_aug_amd import config as CFG, weights as W  # noqa: E402
from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dtype = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
keep = len(sys.argv) > 3 and sys.argv[3] == "1"
dev = torch.device("cuda:0")
cfgs = CFG.tiny()
fam = W.synth_family(cfgs, seed=3)
ids = np.random.RandomState(1).randint(0, cfgs["text"]["vocab"] - 2, (2, 77))
neg = np.random.RandomState(2).randint(0, cfgs["text"]["vocab"] - 2, (1, 77))
ctrl = (np.random.RandomState(3).rand(2, 64, 64, 3) > 0.9).astype(np.uint8) * 255
lat = torch.randn(2, 4, 8, 8)
alive = []
for i in range(n):
    pipe = StableDiffusionControlNetPipeline(fam, cfgs).to(dev, dtype)
    pipe.generate_batch(ids, neg, ctrl, lat, 10)
    torch.cuda.synchronize()
    if keep:
        alive.append(pipe)
    del pipe
    gc.collect()
    if i % 10 == 9:
        free, tot = torch.cuda.mem_get_info()
        print(f"{i + 1} captures: rss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.2f} GB, device in use {(tot - free) / 1e9:.2f} GB, "
              f"torch reserved {torch.cuda.memory_reserved() / 1e9:.2f} GB", flush=True)
print("survived", n)

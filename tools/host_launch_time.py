"""Host-side cost of enqueuing one batch-8 generation (Python + ctypes launches, no sync) against its GPU time."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import config as CFG, ops
from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline
from saspa_aug_amd.synthetic import negative_prompt_ids, synthetic_image, synthetic_prompt_ids
dev = torch.device('cuda:0')
pipe = StableDiffusionControlNetPipeline.from_synthetic(CFG.SD15, 0).to(dev, torch.bfloat16)
b = 8
imgs = torch.from_numpy(np.stack([synthetic_image(512, 512, i) for i in range(b)])).to(dev)
ids = synthetic_prompt_ids(b); neg = negative_prompt_ids()
lat = torch.randn((b, 4, 64, 64), generator=torch.manual_seed(1), dtype=torch.float16)
ctrl = ops.canny(imgs, 120, 200)
pipe.generate_batch(ids, neg, ctrl, lat, 2); torch.cuda.synchronize()
for steps in (50, 50):
    torch.cuda.synchronize(); t0 = time.time()
    out = pipe.generate_batch(ids, neg, ctrl, lat, steps)
    t1 = time.time()
    torch.cuda.synchronize(); t2 = time.time()
    print(f"steps {steps}: host enqueue {t1 - t0:.3f} s, GPU done after {t2 - t0:.3f} s", flush=True)

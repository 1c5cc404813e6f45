"""Which host call blocks while a previous generation is still running on the GPU?  (diagnostics)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import config as CFG, ops
from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline
from saspa_aug_amd.synthetic import negative_prompt_ids, synthetic_image, synthetic_prompt_ids
dev = torch.device('cuda:0')
pipe = StableDiffusionControlNetPipeline.from_synthetic(CFG.SD15, 0).to(dev, torch.bfloat16)
b = 8
imgs_h = np.stack([synthetic_image(512, 512, i) for i in range(b)])
ids = synthetic_prompt_ids(b); neg = negative_prompt_ids()
lat = torch.randn((b, 4, 64, 64), generator=torch.manual_seed(1), dtype=torch.float16)
def gen():
    src = ops.h2d(torch.from_numpy(imgs_h), dev)
    ctrl = ops.canny(src, 120, 200)
    return pipe.generate_batch(ids, neg, ctrl, lat, 50)
gen(); gen(); torch.cuda.synchronize()
T = lambda: time.time()
for trial in range(2):
    out = gen()                                  # GPU busy for ~1.4 s from here; host returns after ~0.8 s
    t0 = T(); x = torch.from_numpy(imgs_h).pin_memory(); t1 = T()
    xd = x.to(dev, non_blocking=True); t2 = T()
    c = ops.canny(xd, 120, 200); t3 = T()
    e = pipe.encode_prompts(ids); t4 = T()
    l = pipe.latents_to_device(lat); t5 = T()
    z = torch.empty((1 << 28,), device=dev, dtype=torch.uint8); t6 = T()
    g = next(iter(pipe._graphs.values()))
    torch.cuda.synchronize(); t7 = T()
    print(f"trial {trial}: pin {t1-t0:.3f} h2d {t2-t1:.3f} canny {t3-t2:.3f} clip {t4-t3:.3f} latents {t5-t4:.3f} empty256MB {t6-t5:.3f} final sync {t7-t6:.3f}", flush=True)

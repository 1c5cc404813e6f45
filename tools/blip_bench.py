"""BASELINE.json configs[2] timing (parity-test configuration, not the bench line): BLIP-Diffusion + Canny ControlNet,
batch 8, 512x512, 50 PLMS steps (51 UNet+ControlNet evaluations), subject front-end included, bf16, synthetic weights."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import config as CFG, ops
from saspa_aug_amd.pipeline import BlipDiffusionControlNetPipeline
from saspa_aug_amd.synthetic import negative_prompt_ids, synthetic_image, synthetic_prompt_ids
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pipe = BlipDiffusionControlNetPipeline.from_synthetic(CFG.BLIP_DIFFUSION, 0).to(dev, torch.bfloat16)
b = 8
imgs = torch.from_numpy(np.stack([synthetic_image(512, 512, i) for i in range(b)])).to(dev)
subjects = [synthetic_image(400, 360, 100 + i) for i in range(b)]
ids = synthetic_prompt_ids(b)[:, :61].copy(); ids[:, -1] = 49407
neg = negative_prompt_ids()
lat = torch.randn((b, 4, 64, 64), generator=torch.manual_seed(1), dtype=torch.float16)

def once():
    ctrl = ops.canny(imgs, 120, 200)
    q = pipe.get_query_embeddings(subjects, ["bird"] * b)
    return pipe.generate_batch(ids, neg, ctrl, lat, steps, 7.5, 1.0, query_embeds=q)

out = once(); torch.cuda.synchronize()
t0 = time.time(); n = 2
for _ in range(n): out = once()
torch.cuda.synchronize(); dt = (time.time() - t0) / n
out2 = once()
torch.cuda.synchronize()
t1 = time.time(); q = pipe.get_query_embeddings(subjects, ["bird"] * b); torch.cuda.synchronize(); tq = time.time() - t1
print(json.dumps({"workload": "BLIP-Diffusion + Canny ControlNet, batch=8 512x512, %d PLMS steps (BASELINE configs[2])" % steps,
                  "images_per_s": round(b / dt, 4), "s_per_batch": round(dt, 3), "front_end_ms_per_batch": round(tq * 1e3, 1),
                  "deterministic": bool(torch.equal(out, out2)), "finite": bool(out.float().isfinite().all()),
                  "dtype": "bf16", "data": "synthetic"}))

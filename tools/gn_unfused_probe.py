"""Which GroupNorms of a batch-8 512x512 generation still run a statistics pass (their input carries no epilogue statistics)?
Patches the library proxy to log every saspa_groupnorm_stats launch with the tensor shape; 2 DDIM steps, eager.
usage: python tools/gn_unfused_probe.py"""
import collections, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ["SASPA_GRAPH"] = "0"
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import config as CFG, ops
from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline
from saspa_aug_amd.synthetic import negative_prompt_ids, synthetic_image, synthetic_prompt_ids
dev = torch.device('cuda:0')
pipe = StableDiffusionControlNetPipeline.from_synthetic(CFG.SD15, 0).to(dev, torch.bfloat16)
seen = collections.Counter()
real = ops.groupnorm
def logged(x, gamma, beta, groups, eps, act=ops.ACT_NONE, x2=None, out=None):
    g = getattr(x, "saspa_gn", None)
    fresh = g is not None and g[2] == x.data_ptr() and g[3] == x._version
    g2 = getattr(x2, "saspa_gn", None) if x2 is not None else None
    fresh2 = x2 is None or (g2 is not None and g2[2] == x2.data_ptr() and g2[3] == x2._version)
    seen[(tuple(x.shape), None if x2 is None else x2.shape[3], "epilogue statistics" if (fresh and fresh2) else "STATISTICS PASS")] += 1
    return real(x, gamma, beta, groups, eps, act, x2, out)
ops.groupnorm = logged
import saspa_aug_amd.models as M
b = 8
imgs = torch.from_numpy(np.stack([synthetic_image(512, 512, i) for i in range(b)])).to(dev)
lat = pipe.latents_to_device(torch.randn((b, 4, 64, 64), generator=torch.manual_seed(1), dtype=torch.float16))
ctrl = ops.canny(imgs, 120, 200)
pipe.generate_batch(synthetic_prompt_ids(b), negative_prompt_ids(), ctrl, lat, 2, 7.5, 0.75, latents_on_device=True)
torch.cuda.synchronize()
for k, v in sorted(seen.items(), key=lambda kv: (kv[0][2], -kv[1])):
    print(f"{v:4d} x  {k[2]:20s} x {k[0]}{'' if k[1] is None else ' + ' + str(k[1])}")

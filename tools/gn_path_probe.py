"""Which GroupNorm call sites of one UNet + ControlNet evaluation still run a statistics pass?  (round 6 diagnostic)
Wraps ops.groupnorm for one eager batch-8 512x512 evaluation and prints, per call, the input geometry, whether each source
carries fresh epilogue statistics (conv(..., gn_unit=...)) and which path ran: fused (no statistics pass) / onepass / stats+apply."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import config as CFG, ops
from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline
from saspa_aug_amd.synthetic import negative_prompt_ids, synthetic_image, synthetic_prompt_ids
os.environ["SASPA_GRAPH"] = "0"
dev = torch.device('cuda:0')
res = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)
pipe = StableDiffusionControlNetPipeline.from_synthetic(CFG.SD15, 0).to(dev, torch.bfloat16)
pipe.safety_checker = None
b = 8
ids, neg = synthetic_prompt_ids(b), negative_prompt_ids()
imgs = torch.from_numpy(np.stack([synthetic_image(res[0], res[1], i) for i in range(b)])).to(dev)
lat = pipe.latents_to_device(torch.randn((b, 4, res[0] // 8, res[1] // 8), generator=torch.manual_seed(1), dtype=torch.float16))
ctrl = ops.canny(imgs, 120, 200)
pipe.generate_batch(ids, neg, ctrl, lat, 1, 7.5, 0.75, latents_on_device=True)
log = []
real = ops.groupnorm
stats_calls = [0]
real_stats = None


def fresh(t):
    g = getattr(t, "saspa_gn", None) if t is not None else None
    if g is None:
        return "none"
    if g[2] != t.data_ptr():
        return "stale-ptr"
    if g[3] != t._version:
        return "stale-version"
    if g[0].shape[0] * 128 != t.shape[0] * t.shape[1] * t.shape[2]:
        return "blocks"
    return f"ok(unit {g[1]})"


def spy(x, gamma, beta, groups, eps, act=0, x2=None, out=None):
    log.append((tuple(x.shape), None if x2 is None else x2.shape[3], fresh(x), fresh(x2) if x2 is not None else "-"))
    return real(x, gamma, beta, groups, eps, act, x2=x2, out=out)


ops.groupnorm = spy
import saspa_aug_amd.models as M
rec_kinds = collections.Counter()


class Rec:
    def __call__(self, kind, flops, call, meta=None):
        rec_kinds[kind] += 1
        return call()


pipe.generate_batch(ids, neg, ctrl, lat, 1, 7.5, 0.75, latents_on_device=True)
n_eval = len(log)
ops.set_recorder(Rec())
log2 = log[:]
log.clear()
pipe.generate_batch(ids, neg, ctrl, lat, 1, 7.5, 0.75, latents_on_device=True)
ops.set_recorder(None)
print("launch kinds of one generation (1 step):", {k: v for k, v in rec_kinds.items() if "groupnorm" in k or "gn" in k})
agg = collections.Counter()
for e in log2:
    agg[e] += 1
for (shape, c1, f0, f1), n in sorted(agg.items(), key=lambda kv: (-kv[0][0][1], str(kv[0]))):
    print(f"x {str(shape):24s} x2 channels {str(c1):5s} stats(x) {f0:16s} stats(x2) {f1:16s} x{n}")

"""Per-shape time / TFLOP/s table of the implicit-GEMM and flash-attention launches of one
UNet+ControlNet evaluation at the bench shape (B=8 images, CFG -> 16 samples, 512x512).
usage: python tools/gemm_shapes.py [H W]   (e.g. 512 704: the dominant FGVC-Aircraft bucket, BASELINE configs[3])"""
import sys, collections
import numpy as np, torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import saspa_aug_amd
from saspa_aug_amd import config as CFG, ops
from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline
from saspa_aug_amd.synthetic import negative_prompt_ids, synthetic_image, synthetic_prompt_ids
from bench import Recorder
dev = torch.device('cuda:0')
pipe = StableDiffusionControlNetPipeline.from_synthetic(CFG.SD15, 0).to(dev, torch.bfloat16)
b = 8
HH, WW = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)
imgs = torch.from_numpy(np.stack([synthetic_image(HH, WW, i) for i in range(b)])).to(dev)
ids = synthetic_prompt_ids(b); neg = negative_prompt_ids()
lat = torch.randn((b, 4, HH // 8, WW // 8), generator=torch.manual_seed(1), dtype=torch.float16)
ctrl = ops.canny(imgs, 120, 200)
pipe.generate_batch(ids, neg, ctrl, lat, 1)
rec = Recorder(); ops.set_recorder(rec)
pipe.generate_batch(ids, neg, ctrl, lat, 3)
ops.set_recorder(None); torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, flops, e0, e1, meta in rec.items:
    d = agg.setdefault((kind, meta), [0, 0.0, 0.0]); d[0] += 1; d[1] += flops; d[2] += e0.elapsed_time(e1)
tot = sum(v[2] for v in agg.values())
print(f"total recorded ms {tot:.1f}")
rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
print(f"{'kind':10s} {'meta (M,N,K,kh,stride,up,concat)|(B,h,nq,nk,d)':55s} {'n':>5s} {'ms':>9s} {'%':>6s} {'TF/s':>8s} {'us/launch':>10s}")
for (kind, meta), (n, fl, ms) in rows[:70]:
    print(f"{kind:10s} {str(meta):55s} {n:5d} {ms:9.2f} {100*ms/tot:6.2f} {fl/ms/1e9:8.1f} {1e3*ms/n:10.1f}")

"""GPU rate of the sampling path (Canny + CLIP + 50 x (UNet + ControlNet) + VAE + safety checker, batch 8, bf16, hipGraph) at
the image sizes of BASELINE configs[3]: `resize_image` (all_utils/utils.py:58-79) turns FGVC-Aircraft photographs into 512x704
(most), 512x768 and 512x512.  Prints images/s per size and the rate FLOP scaling of the 512x512 result predicts (conv /
linear terms x HW / 512^2, self-attention x (HW / 512^2)^2: SURVEY 8d).  usage: python tools/nonsquare_bench.py [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import config as CFG, ops
from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline
from saspa_aug_amd.synthetic import negative_prompt_ids, synthetic_image, synthetic_prompt_ids
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pipe = StableDiffusionControlNetPipeline.from_synthetic(CFG.SD15, 0).to(dev, torch.bfloat16)
b = 8
ids, neg = synthetic_prompt_ids(b), negative_prompt_ids()
base = None
for (hh, ww) in ((512, 512), (512, 704), (512, 768)):
    imgs = torch.from_numpy(np.stack([synthetic_image(hh, ww, i) for i in range(b)])).to(dev)
    lat = pipe.latents_to_device(torch.randn((b, 4, hh // 8, ww // 8), generator=torch.manual_seed(1), dtype=torch.float16))
    def run():
        ctrl = ops.canny(imgs, 120, 200)
        return pipe.generate_batch(ids, neg, ctrl, lat, steps, 7.5, 0.75, latents_on_device=True)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.time(); run(); torch.cuda.synchronize(); ts.append(time.time() - t0)
    dt = sorted(ts)[1]
    r = hh * ww / 512.0 ** 2
    # per image 2135.06 GFLOP per step of which 16.5 % attention matmuls (self-attention dominates) + 2579 fixed
    flops = steps * 2135.06 * (0.835 * r + 0.165 * r * r) + 2579.2 * r
    rate = b / dt
    if base is None:
        base = (rate, flops)
    print(f"{hh}x{ww}: {rate:6.3f} images/s  ({dt * 1e3 / steps:6.2f} ms per step incl. fixed work), {flops / 1e3:6.1f} TFLOP/image -> "
          f"{rate * flops / 1e3:6.1f} TFLOP/s; FLOP-scaled from 512x512: {base[0] * base[1] / flops:6.3f} images/s", flush=True)

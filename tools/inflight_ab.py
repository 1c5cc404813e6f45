"""Two generations in flight: does the chip deliver more images/s when two independent batch-8 generations (each its own
pipeline object, step graph, static buffers and HIP stream, driven by its own host thread) run side by side than when the
same batches run one after the other?  Same kernels, same dispatch -- what changes is that one generation's launch gaps, tile
tails, store-bound epilogues and small deep-level kernels are covered by the other's work (the mechanism behind the two-branch
step graph, extended to the whole step incl. the single-branch UNet decoder, VAE, CLIP and Canny).
usage: python tools/inflight_ab.py [batches per arm = 6] [res = 512] [ddim steps = 50]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import saspa_aug_amd  # noqa: F401,E402
from bench import ClockSampler  # noqa: E402
from saspa_aug_amd import config as CFG  # noqa: E402
from saspa_aug_amd import ops  # noqa: E402
from saspa_aug_amd import weights as W  # noqa: E402
from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline  # noqa: E402
from saspa_aug_amd.synthetic import negative_prompt_ids, synthetic_image, synthetic_prompt_ids  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
RES = int(sys.argv[2]) if len(sys.argv) > 2 else 512
S = int(sys.argv[3]) if len(sys.argv) > 3 else 50
B = 8
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfgs = CFG.SD15
fam = W.synth_family(cfgs, 0)
pipes = [StableDiffusionControlNetPipeline(dict(fam), cfgs).to(dev, torch.bfloat16) for _ in range(2)]
del fam
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
vocab = cfgs["text"]["vocab"]
neg = negative_prompt_ids(vocab)


def make_batch(pipe, idx):
    base = idx * B
    imgs = np.stack([synthetic_image(RES, RES, base + i) for i in range(B)])
    ids = synthetic_prompt_ids(B, seed=1 + base, vocab=vocab)
    lat = torch.randn((B, 4, RES // 8, RES // 8), generator=torch.Generator().manual_seed(1 + base), dtype=torch.float16)
    return torch.from_numpy(imgs).to(dev), torch.from_numpy(ids).to(dev), pipe.latents_to_device(lat)


def hot_path(pipe, batch):
    imgs, ids, lat = batch
    ctrl = ops.canny(imgs, 120, 200)
    return pipe.generate_batch(ids, neg, ctrl, lat, S, 7.5, 0.75, latents_on_device=True)


batches = [make_batch(pipes[0], i) for i in range(2 * N)]
outs = {}
for k in range(2):                                   # warm + capture, one pipeline at a time (capture is process-global)
    with torch.cuda.stream(streams[k]):
        outs[k] = hot_path(pipes[k], batches[k]).clone()
        hot_path(pipes[k], batches[k])
    torch.cuda.synchronize()


def worker(k, idxs, res):
    torch.cuda.set_device(0)
    with torch.cuda.stream(streams[k]):
        for i in idxs:
            res[i] = hot_path(pipes[k], batches[i])
    streams[k].synchronize()


def arm(two):
    res = {}
    torch.cuda.synchronize()
    cs = ClockSampler(dev).start()
    t0 = time.time()
    if two:
        th = [threading.Thread(target=worker, args=(k, list(range(k, 2 * N, 2)), res)) for k in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
    else:
        worker(0, list(range(2 * N)), res)
    torch.cuda.synchronize()
    dt = time.time() - t0
    clk = cs.stop()
    return 2 * N * B / dt, clk, res


for rnd in range(2):
    v1, c1, r1 = arm(False)
    v2, c2, r2 = arm(True)
    same = all(torch.equal(r1[i], r2[i]) for i in range(0, 2 * N, 2))          # pipeline 0 ran the even batches in both arms
    print(f"round {rnd}: one at a time {v1:.3f} images/s (sclk {c1 and c1['sclk_mhz_mean']}),  two in flight {v2:.3f} images/s "
          f"(sclk {c2 and c2['sclk_mhz_mean']})  ratio {v2 / v1:.4f}  even batches bit-identical: {same}", flush=True)

"""Host cost of ONE hipGraph replay of the captured sampling step, separated from back-pressure: after a device synchronise the
first replays go into an empty queue, so the time each `replay()` call blocks the host is the runtime's own enqueue cost
(ROCm walks the graph's kernel nodes one by one); once the queue is full the call time tracks the GPU.
usage: python tools/graph_host_cost.py"""
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
pipe.generate_batch(ids, neg, ctrl, lat, 50); torch.cuda.synchronize()
g = [v for v in pipe._graphs.values() if v.graph is not None][-1]
for trial in range(3):
    g.idx.zero_()                      # the step counter indexes the per-step tables: never replay past the 50 captured steps
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for i in range(12):
        a = time.perf_counter()
        g.graph.replay()
        ts.append((time.perf_counter() - a) * 1e3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"trial {trial}: per-replay host ms {['%.1f' % t for t in ts]}; 12 replays: host {1e3 * (t1 - t0):.0f} ms, GPU done after {1e3 * (t2 - t0):.0f} ms "
          f"({1e3 * (t2 - t0) / 12:.1f} ms per step)", flush=True)

"""Soak of the product's DEFAULT capture form (two-branch step graphs, SASPA_FORK=1) in a FRESH child process: many pipelines
created and destroyed, several (H, W) buckets each, every graph replayed for several generations -- the session shape in which
ROCm 7.2's hipGraphLaunch ended in a segmentation fault (hip::Graph::UpdateStreams, profiles/r5_graph_replay_segv_backtrace.txt)
while each pipeline took its own capture side stream.  With the shared side stream (pipeline.side_stream, round 6) the child
must exit cleanly and every generation must reproduce its first result bit for bit.  (Advisor finding, round 5.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import gc, os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
import saspa_aug_amd
from saspa_aug_amd import config as CFG, weights as W
from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline, fork_enabled
from saspa_aug_amd.synthetic import synthetic_image
from saspa_aug_amd import ops
assert fork_enabled(), "the soak is about the two-branch form"
dev = torch.device("cuda:0")
cfgs = CFG.tiny()
sizes = [(64, 64), (64, 128), (128, 64), (128, 128)]
keep, n_gen = [], 0
for k in range(%(pipes)d):
    pipe = StableDiffusionControlNetPipeline(W.synth_family(cfgs, seed=k %% 3), cfgs).to(dev, torch.bfloat16 if k %% 2 else torch.float32)
    for (h, w) in sizes[: 2 + k %% 3]:
        img = np.stack([synthetic_image(h, w, i) for i in range(2)])
        ctrl = ops.canny(torch.from_numpy(img).to(dev), 120, 200)
        ids = np.random.RandomState(k).randint(0, 500, (2, 77))
        neg = np.random.RandomState(99).randint(0, 500, (1, 77))
        lat = torch.randn((2, 4, h // 8, w // 8), generator=torch.manual_seed(k))
        first = pipe.generate_batch(ids, neg, ctrl, lat, 4).clone()
        for _ in range(%(replays)d):
            again = pipe.generate_batch(ids, neg, ctrl, lat, 4)
            assert torch.equal(first, again), (k, h, w)
            n_gen += 1
    if k %% 4 == 0:
        keep.append(pipe)            # some pipelines (and their graphs) stay alive next to the new ones
    else:
        del pipe
        gc.collect()
        torch.cuda.empty_cache()
torch.cuda.synchronize()
print(f"soak ok: %(pipes)d pipelines, {n_gen} replayed generations, {len(keep)} kept alive", flush=True)
"""


@pytest.mark.gpu
def test_two_branch_graphs_soak_in_a_fresh_process():
    env = dict(os.environ, SASPA_FORK="1")
    env.pop("SASPA_SIDE_STREAM", None)
    code = CHILD % dict(root=ROOT, pipes=int(os.environ.get("SASPA_SOAK_PIPES", "24")), replays=3)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, f"child exited with {r.returncode} (negative = signal)\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    assert "soak ok" in r.stdout

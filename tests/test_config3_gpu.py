"""BASELINE configs[3] in miniature: ONE rank's shard of a multi-rank plan with mixed (H, W) buckets, end to end through
saspa_aug_amd.run_aug.main on the device (PNG writer processes on), checked against the oracle for two items of different
sizes.  The full-size rehearsal (3 334 images x 4 variants, world 8) is tools/config3_rehearsal.py ->
profiles/r5_config3_rehearsal.json; the real collective is tests/test_distributed_gloo.py."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

import saspa_aug_amd  # noqa: F401
from oracle import canny as OC
from oracle import pipeline as OP
from saspa_aug_amd import config as CFG
from saspa_aug_amd import run_aug as R
from saspa_aug_amd import weights as W
from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline

pytestmark = pytest.mark.gpu


class _LocalWorld:
    """torch.distributed's surface as run_aug.main uses it, at (rank 0, world n) in one process."""

    def __init__(self, world):
        self.world = world

    def get_rank(self):
        return 0

    def get_world_size(self):
        return self.world

    def get_backend(self):
        return "gloo"

    def barrier(self):
        pass

    def gather(self, t, gathered, dst=0):
        gathered[0].copy_(t)
        for g in gathered[1:]:
            g.zero_()


def test_mixed_bucket_shard_matches_oracle(dev, tmp_path):
    cfgs = CFG.tiny()
    fam = W.synth_family(cfgs, seed=3)
    pipe = StableDiffusionControlNetPipeline(fam, cfgs).to("cuda:0", torch.float32)
    prompts = tmp_path / "prompts.txt"
    prompts.write_text("".join(f"an airplane in scene {k}.\n" for k in range(6)))
    sizes = ((64, 64), (64, 128), (128, 64), (64, 192))
    steps = 3
    s = R.Settings(DATASET="synthetic", BASE_MODEL="sd_v1.5", RESOLUTION=64, NUM_INFERENCE_STEPS=steps, NUM_PER_IMAGE=2, SEED=1,
                   SEMANTIC_FILTERING=0, MODEL_CONFIDENCE_BASED_FILTERING=0, PROMPTS_FILE=str(prompts), BATCH_SIZE=4, PRECISION="fp32",
                   DATASET_KWARGS=dict(root_path=str(tmp_path / "ds" / "data"), n_images=14, sizes=sizes))
    res = R.main(s, pipe=pipe, dist=_LocalWorld(2))
    items, mine = res["items"], res["mine"]
    assert len(items) == 28 and 0 < len(mine) < len(items)
    # contiguous shard, balanced by area, with at least three sizes in it -> several buckets, ragged tail batches
    assert [it.order for it in mine] == list(range(mine[0].order, mine[-1].order + 1)) and mine[0].order == 0
    area = sum(it.height * it.width for it in items)
    assert abs(sum(it.height * it.width for it in mine) / area - 0.5) < 0.08
    assert len({(it.height, it.width) for it in mine}) >= 3
    assert res["n_batches"] > len(mine) // s.BATCH_SIZE          # buckets do not fill whole batches
    mine_orders = {it.order for it in mine}
    for it in items:
        assert (res["status"][it.order].item() == 1) == (it.order in mine_orders)
        assert Path(it.output_path).exists() == (it.order in mine_orders)
    assert res["png_submitted"] >= len(mine)
    body = json.load(open(res["json_path"]))
    assert len(body) == 14 and sum(len(v) for v in body.values()) == len(mine)
    # two items of different buckets against the oracle: same prompt tokens, control image, slice of the noise stream
    tok = pipe.tokenizer
    neg = torch.from_numpy(tok(R.NEGATIVE_PROMPT))
    picks, seen = [], set()
    for it in mine:
        if (it.height, it.width) not in seen and it.i == 1:      # second variants: their noise sits mid-stream
            seen.add((it.height, it.width))
            picks.append(it)
    picks = picks[:2]
    assert len(picks) == 2 and (picks[0].height, picks[0].width) != (picks[1].height, picks[1].width)
    noises = R.noise_for_items(items, picks, s.SEED, pipe.noise_dtype)
    for it in picks:
        src = R.load_raw(it.source_path)
        assert src.shape[:2] == (it.height, it.width)
        ctrl = OC.generate_canny_array(src, s.LOW_THRESHOLD_CANNY, s.HIGH_THRESHOLD_CANNY)
        ref = OP.sd_controlnet_pipeline(fam, cfgs, torch.from_numpy(tok(it.prompt)), neg, ctrl, noises[it.order].float(), steps,
                                        s.GUIDANCE_SCALE, s.CONTROLNET_CONDITIONING_SCALE)
        got = np.asarray(Image.open(it.output_path).convert("RGB"))
        ref = np.asarray(ref)[0]
        assert got.shape == ref.shape == (it.height, it.width, 3)
        assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1, f"item {it.order} ({it.height}x{it.width}) deviates from the oracle"

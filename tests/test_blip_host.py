"""CPU tests of the BLIP-Diffusion host logic: PNDM plan vs the oracle's step_plms, prompt amplification,
tokenizers.  (No GPU, no kernels.)"""
import numpy as np
import torch

import saspa_aug_amd  # noqa: F401
from oracle import pipeline as OP
from saspa_aug_amd.pipeline import BlipDiffusionControlNetPipeline
from saspa_aug_amd.scheduler import PNDMScheduler
from saspa_aug_amd.tokenizer import BertHashTokenizer, BertWordPieceTokenizer, HashTokenizer


def test_pndm_plan_replays_step_plms():
    for steps in (1, 2, 3, 4, 5, 9, 30):
        o = OP.PNDM()
        ts = o.set_timesteps(steps)
        plan = PNDMScheduler().plan(steps)
        assert len(plan) == (steps + 1 if steps > 1 else 1)
        assert [t for t, _ in plan] == [int(t) for t in ts]
        g = torch.Generator().manual_seed(steps)
        x = torch.randn(3, 5, generator=g)
        xr = x.clone()
        hist, saved = [torch.zeros_like(x) for _ in range(4)], None
        for (t, d), tt in zip(plan, ts):
            e = torch.randn(3, 5, generator=g)
            xr = o.step(e, tt, xr)
            s = saved if d["use_saved"] else x
            if d["save_sample"]:
                saved = x.clone()
            m = d["w_cur"] * e + sum(w * h for w, h in zip(d["w_hist"], hist))
            assert d["store_slot"] < 0 or d["w_hist"][d["store_slot"]] == 0.0     # never combine the slot being overwritten
            if d["store_slot"] >= 0:
                hist[d["store_slot"]] = e
            x = d["coef_sample"] * s + d["coef_model"] * m
        assert (x - xr).abs().max().item() < 2e-5


def test_pndm_timesteps_sd15_50_steps():
    ts = PNDMScheduler().set_timesteps(50)
    assert len(ts) == 51 and ts[0] == 981 and ts[1] == 961 and ts[2] == 961 and ts[-1] == 1


def test_build_prompt_and_token_budget():
    p = BlipDiffusionControlNetPipeline.build_prompt(" on a snowy field ", "bird")
    assert p.startswith("a bird on a snowy field, a bird on a snowy field") and p.count("a bird on a snowy field") == 20
    ids = HashTokenizer()(p, max_len=61)
    assert ids.shape == (1, 61) and ids[0, 0] == 49406 and ids[0, -1] == 49407


def test_bert_tokenizers(tmp_path):
    assert BertHashTokenizer()("Bird")[0, 0] == 101 and BertHashTokenizer()("bird")[0, -1] == 102
    assert np.array_equal(BertHashTokenizer()("Bird"), BertHashTokenizer()("bird"))
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "bird", "air", "##plane", "##s", "a", ","]
    f = tmp_path / "vocab.txt"
    f.write_text("\n".join(vocab) + "\n")
    t = BertWordPieceTokenizer(str(f))
    assert t("Airplanes, a bird").tolist() == [[2, 5, 6, 7, 9, 8, 4, 3]]
    assert t("zebra").tolist() == [[2, 1, 3]]

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


def test_blip_plan_replays_subject_choice(tmp_path):
    """run_aug/run_aug.py:382 (numpy prompt draw per image) then, per NOT-skipped variant, :446
    random.choice(same-class paths) on the python stream; artistic prompts are off for BLIP (:529)."""
    import random
    from pathlib import Path

    from saspa_aug_amd import run_aug as R
    from saspa_aug_amd import utils as U
    s = R.Settings(DATASET="planes", BASE_MODEL="blip_diffusion", USE_ARTISTIC_PROMPTS=False, NUM_PER_IMAGE=2, SEED=3)
    assert R.prompt_str_for(s) == "gpt-meta_class_prompt_w_sub_class_style_img_from_diff_img"
    assert R.output_folder_for(s, "root").startswith("root/aug_data/controlnet/blip_diffusion/canny/")
    paths = [f"/d/{n}.jpg" for n in "abcd"]
    classes = {"a": "X", "b": "Y", "c": "X", "d": "X"}
    same = lambda p: [q for q in paths if classes[Path(q).stem] == classes[Path(p).stem]]   # noqa: E731
    prompts = [f"an airplane {k}" for k in range(7)]
    U.set_seed(3)
    items = R.plan_work(s, paths, prompts, str(tmp_path), classes, image_size_fn=lambda p: (512, 512), same_class_fn=same)
    py_after, np_after = random.random(), float(np.random.rand())
    # hand replay
    U.set_seed(3)
    exp = []
    for p in paths:
        for pr in np.random.choice(prompts, 2):
            exp.append((str(pr).replace("airplane", f"{classes[Path(p).stem]} airplane"), random.choice(same(p))))
    assert [(it.prompt, it.subject_path) for it in items] == exp
    assert (random.random(), float(np.random.rand())) == (py_after, np_after)
    # an existing output is skipped BEFORE the subject draw: later picks shift on the python stream
    Path(items[2].output_path).touch()
    U.set_seed(3)
    again = R.plan_work(s, paths, prompts, str(tmp_path), classes, image_size_fn=lambda p: (512, 512), same_class_fn=same)
    assert again[2].skip and again[2].subject_path is None
    U.set_seed(3)
    picks = []
    for k, p in enumerate(paths):
        np.random.choice(prompts, 2)
        for i in range(2):
            picks.append(None if 2 * k + i == 2 else random.choice(same(p)))
    assert [it.subject_path for it in again] == picks


def test_blip_call_kwargs_follow_the_reference():
    from PIL import Image

    from saspa_aug_amd import run_aug as R
    seen = {}

    class FakePipe:
        def __call__(self, **kw):
            seen.update(kw)
            return type("O", (), {"images": ["img"]})()
    ctrl, subj = Image.new("RGB", (96, 64)), Image.new("RGB", (50, 40))
    out = R.pass_thorugh_pipe("blip_diffusion", FakePipe(), "a prompt", subj, 0, 0.85, 30, None, 7.5, 0.75, control_image=ctrl,
                              blip_src_category="bird", blip_target_category="bird")
    assert out == "img"
    assert seen["reference_image"] is subj and seen["condtioning_image"] is ctrl and (seen["height"], seen["width"]) == (64, 96)
    assert seen["neg_prompt"] == R.NEGATIVE_PROMPT and "negative_prompt" not in seen and "controlnet_conditioning_scale" not in seen
    assert seen["source_subject_category"] == seen["target_subject_category"] == "bird"

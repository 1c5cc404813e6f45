"""Round-4 host-side additions (CPU): the baseline prompt sources of run_aug/run_aug.py:303-339 / :589-666 (ALIA, txt2sentence,
txt2sentence-per_class), replayed by hand against the reference's loop arithmetic; the twin-branch dispatch hint of ops."""
import json
import random

import numpy as np

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
from saspa_aug_amd import run_aug as R
from saspa_aug_amd import utils


def test_read_prompts_from_json_matches_reference_reader(tmp_path):
    """prompts_engineering/blip_utils.py:14-27: per_class -> the dict, else the concatenation of its values in file order."""
    d = {"A": ["a plane on a runway.", "x" * 200], "B": ["a plane in the sky"]}
    f = tmp_path / "p.json"
    f.write_text(json.dumps(d))
    flat = R.read_prompts_from_json(str(f), "planes", per_class=False)
    assert flat == ["a plane on a runway.", "x" * R.MAX_PROMPT_LENGTH, "a plane in the sky"]
    per = R.read_prompts_from_json(str(f), "planes", per_class=True)
    assert list(per) == ["A", "B"] and per["A"][1] == "x" * R.MAX_PROMPT_LENGTH


def test_default_prompt_files_of_the_baseline_sources():
    for ds in ("planes", "cars", "cub", "dtd", "compcars-parts", "planes_biased"):
        f = R.default_prompts_file(R.Settings(DATASET=ds, PROMPT_TYPE="ALIA"))
        assert f.endswith(f"ALIA_prompts/gpt_output/{ds}_prompts.txt") and len(R.read_prompts(f)) > 0
    f = R.default_prompts_file(R.Settings(DATASET="cars", PROMPT_TYPE="txt2sentence"))
    assert f.endswith("LE_200_cars_all_classes_False.json") and len(R.read_prompts_from_json(f)) > 100
    try:                                                   # the reference ships no planes file: explicit, not a silent default
        R.default_prompts_file(R.Settings(DATASET="planes", PROMPT_TYPE="txt2sentence-per_class"))
        assert False
    except NotImplementedError as e:
        assert "PROMPTS_FILE" in str(e)


def test_plan_per_class_prompts_replays_the_reference_loop(tmp_path):
    """run_aug/run_aug.py:365-367 + :380-382: with txt2sentence-per_class the image's own class's prompt list (trailing '.'
    stripped) is what np.random.choice draws from, keyed by the image stem for planes / cars."""
    s = R.Settings(DATASET="planes", PROMPT_TYPE="txt2sentence-per_class", NUM_PER_IMAGE=2, SEED=3, USE_ARTISTIC_PROMPTS=False,
                   PROMPT_WITH_SUB_CLASS=False)
    c2p = {"707": ["an airplane at dawn.", "an airplane at dusk", "an airplane at noon"], "A320": ["an airplane parked.", "an airplane landing"]}
    classes = {"img0": "707", "img1": "A320"}
    utils.set_seed(s.SEED)
    items = R.plan_work(s, ["/d/img0.jpg", "/d/img1.jpg"], None, str(tmp_path), classes, image_size_fn=lambda p: (512, 512),
                        class_to_prompts=c2p)
    # hand replay of the reference's draws
    utils.set_seed(s.SEED)
    want = []
    for stem in ("img0", "img1"):
        ps = [q[:-1] if q[-1] == "." else q for q in c2p[classes[stem]]]
        want += [str(q) for q in np.random.choice(ps, 2)]
    assert [it.prompt for it in items] == want


def test_twin_branch_sets_and_restores_the_sharing_hint(monkeypatch):
    assert ops._TWIN[0] is False
    with ops.twin_branch():
        assert ops._TWIN[0] is True
        with ops.twin_branch(False):
            assert ops._TWIN[0] is False
        assert ops._TWIN[0] is True
    assert ops._TWIN[0] is False
    monkeypatch.setenv("SASPA_TWIN_KS", "0")
    with ops.twin_branch():
        assert ops._TWIN[0] is False


def test_xattn_weight_packing_is_a_permutation():
    """weights.pack_xattn_w: rows of to_q and K columns of to_out are permuted, never dropped or duplicated, and the layout
    statements of include/saspa_hip.h hold (head blocks first, the 8-channel tails last)."""
    import torch
    from saspa_aug_amd import weights as W
    c = W.XATTN_C
    wq = torch.arange(c, dtype=torch.float32)[:, None].repeat(1, c)            # row r holds r
    wo = torch.arange(c, dtype=torch.float32)[None, :].repeat(c, 1)            # column k holds k
    w, b = W.pack_xattn_w(wq, wo, torch.arange(c, dtype=torch.float32))
    assert tuple(w.shape) == (2 * c, c) and tuple(b.shape) == (2 * c,)
    rows = w[:c, 0].long().tolist()
    assert sorted(rows) == list(range(c))
    assert rows[:32] == list(range(32)) and rows[32:64] == list(range(40, 72))           # heads 0, 1: channels 0..31
    assert rows[256:264] == list(range(32, 40)) and rows[264:272] == list(range(72, 80))  # tails of heads 0, 1
    cols = w[c, :].long().tolist()
    assert sorted(cols) == list(range(c))
    assert sorted(cols[:32]) == list(range(32))                                       # first two K-steps = head 0, channels 0..31
    assert sorted(cols[256:272]) == list(range(32, 40)) + list(range(72, 80))         # K-step 16 = tails of heads 0 and 1
    assert b[:c].abs().sum() == 0 and b[c:].tolist() == list(range(c))

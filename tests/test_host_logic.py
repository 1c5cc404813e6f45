"""CPU tests of the host side of the path against golden vectors captured from the
reference's own importable functions (tests/golden/make_golden.py ->
reference_host_golden.json): JSON name / body, resize target sizes, HWC3, prompt constants,
RNG replay of the generation loop, the downstream consumer's view of our JSON, sharding and
noise-stream slicing."""
import json
import os
import random
from pathlib import Path

import numpy as np
import pytest
import torch
from PIL import Image

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import run_aug as R
from saspa_aug_amd import utils as U
from saspa_aug_amd.prompts_engineering import ARTISTIC_PROMPTS, IMAGE_VARIATIONS_PROMPTS

GOLD = json.load(open(Path(__file__).parent / "golden" / "reference_host_golden.json"))


def test_aug_json_path_matches_reference():
    folder = str(Path(GOLD["aug_json_path"][0]["path"]).parent / "images")
    for case in GOLD["aug_json_path"]:
        assert U.get_aug_json_path(folder, **case["kwargs"]) == case["path"]
    # the name hard-wired in fgvc/trainings_scripts/consecutive_runs_aug.sh:10
    assert U.get_aug_json_path(folder, semantic_filtering=1, model_confidence_based_filtering=1).endswith(
        "semantic_filtering-model_confidence_based_filtering_top_10_classes-aug.json")


def test_resize_target_sizes_match_reference():
    for c in GOLD["resize_targets"]:
        th, tw, k = U.resize_target_size(c["h"], c["w"], c["res"])
        assert (th, tw) == (c["out_h"], c["out_w"]), c
        assert ("LANCZOS4" if k > 1 else "AREA") == c["interp"], c
    sq = np.random.RandomState(0).randint(0, 255, (512, 512, 3)).astype(np.uint8)
    assert U.resize_image(sq, 512) is sq            # identity for already-sized inputs: no device needed
    import torch
    if not torch.cuda.is_available():               # the resampling itself runs in the gfx950 kernels: no CPU path
        with pytest.raises(RuntimeError):
            U.resize_image(np.zeros((300, 400, 3), np.uint8), 512)


def test_hwc3_matches_reference():
    h = GOLD["hwc3"]
    assert U.HWC3(np.array(h["gray_in"], np.uint8)).tolist() == h["gray_out"]
    assert U.HWC3(np.array(h["rgba_in"], np.uint8)).tolist() == h["rgba_out"]


def test_prompt_constants_match_reference():
    assert ARTISTIC_PROMPTS == GOLD["ARTISTIC_PROMPTS"]
    assert IMAGE_VARIATIONS_PROMPTS == GOLD["IMAGE_VARIATIONS_PROMPTS"]
    assert R.MAX_FILENAME_LENGTH == U.MAX_FILE_NAME_LENGTH == 40


def _settings(**kw):
    return R.Settings(DATASET="planes", **kw)


def test_plan_replays_reference_rng(tmp_path):
    """File names / prompts of a seed-1 run equal the hand replay of run_aug/run_aug.py:380-429,
    and the python / numpy RNG streams end in the same state."""
    g = GOLD["planes_seed1_replay"]
    s = _settings()
    U.set_seed(1)
    paths = [f"/data/images/{st}.jpg" for st in g["stems"]]
    items = R.plan_work(s, paths, GOLD["replay_prompts"], str(tmp_path), g["classes"], image_size_fn=lambda p: (512, 512))
    assert [Path(it.output_path).name for it in items] == g["file_names"]
    assert random.random() == g["py_random_after"] and float(np.random.rand()) == g["np_random_after"]
    assert [it.noise_offset for it in items] == [k * 4 * 64 * 64 for k in range(len(items))]


def test_plan_rng_quirks(tmp_path):
    """(a) odd variants still consume one random.random() at prob 0.5 (short-circuit order);
    (b) prob != 0.5 draws random.random() for every variant; (c) existing outputs are skipped
    before the noise draw so later offsets shift."""
    s = _settings(NUM_PER_IMAGE=3)
    prompts = [f"airplane {k}" for k in range(10)]
    U.set_seed(5)
    a = R.plan_work(s, ["/x/a.jpg", "/x/b.jpg"], prompts, str(tmp_path), {"a": "A", "b": "B"}, image_size_fn=lambda p: (512, 704))
    st = random.getstate()
    random.seed(5)
    for _ in range(2):            # i = 1 of each image
        random.random()
    assert random.getstate() == st
    assert [it.noise_offset for it in a] == [k * 4 * 64 * 88 for k in range(6)]
    # skip-if-exists
    Path(a[1].output_path).touch()
    U.set_seed(5)
    b = R.plan_work(s, ["/x/a.jpg", "/x/b.jpg"], prompts, str(tmp_path), {"a": "A", "b": "B"}, image_size_fn=lambda p: (512, 704))
    assert b[1].skip and b[1].noise_offset == -1 and b[2].noise_offset == a[1].noise_offset
    assert [it.prompt for it in a] == [it.prompt for it in b]
    s2 = _settings(ARTISTIC_PROMPTS_PROB=0.3)
    U.set_seed(5)
    R.plan_work(s2, ["/x/a.jpg"], prompts, str(tmp_path), {"a": "A"}, image_size_fn=lambda p: (512, 512))
    st2 = random.getstate()
    random.seed(5)
    random.random(), random.random()
    assert random.getstate() == st2


def test_output_folder_matches_training_script_path():
    s = _settings()
    root = "data/FGVC-Aircraft/fgvc-aircraft-2013b/data"
    assert R.output_folder_for(s, root) == (root + "/aug_data/controlnet/sd_v1.5/canny/"
                                            "gpt-meta_class_prompt_w_sub_class_artistic_prompts_p_0.5_seed_1/images")


def test_create_json_matches_reference_body(tmp_path, monkeypatch):
    """Our JSON for the fixture folder == the JSON the reference's own
    create_json_of_image_name_to_augmented_images_paths wrote for it (lists compared sorted:
    the reference's order is os.listdir order)."""
    g = GOLD["create_json"]
    monkeypatch.chdir(tmp_path)
    imgs = Path(g["images_dir"])
    imgs.mkdir(parents=True)
    for n in g["listing"]:
        Image.fromarray(np.full((8, 8, 3), 90, np.uint8)).save(imgs / n)
    originals = [f"data/FGVC-Aircraft/fgvc-aircraft-2013b/data/images/{i}.jpg" for i in g["ids"]]
    jp = U.create_json_of_image_name_to_augmented_images_paths("planes", str(imgs), init_log=False,
                                                               original_images_paths=originals)
    assert jp == g["json_path"]
    body = json.load(open(jp))
    assert {k: sorted(v) for k, v in body.items()} == g["body"]
    assert body[f"{g['ids'][4]}.jpg"] == []                 # originals without augmentations keep an empty list


def test_downstream_consumer_contract(tmp_path):
    """What AugWrapperDataset (fgvc/datasets/aug_wrapper_dataset.py:106-150) does with the JSON,
    replayed on ours: drop empty lists, keep the first `limit` entries, random.choice."""
    g, c = GOLD["create_json"], GOLD["consumer"]
    body = g["body"]
    aug = {k: v[:c["limit_aug_per_image"]] for k, v in body.items() if v}
    assert sorted(aug.keys()) == c["kept_keys"]
    files = [f"data/FGVC-Aircraft/fgvc-aircraft-2013b/data/images/{i}.jpg" for i in g["ids"]] * 3
    random.seed(c["seed"])
    picks = []
    for p in files:
        if random.random() < c["aug_sample_ratio"]:
            cand = aug.get(Path(p).name, [p]) or [p]
            p = random.choice(cand)
        picks.append(p)
    assert picks == c["picks"]


def test_shard_and_batches_cover_all_items_once():
    items = [R.WorkItem(k, k // 2, f"/x/{k // 2}.jpg", str(k // 2), k % 2, "p", f"/o/{k}.png", 512, 512 if k % 3 else 704)
             for k in range(37)]
    items[5].skip = True
    for world in (1, 2, 3, 8):
        shards = R.shard_items(items, world)
        flat = [it.order for sh in shards for it in sh]
        assert flat == [it.order for it in items if not it.skip]          # contiguous, ordered, disjoint
        load = [sum(it.height * it.width for it in sh) for sh in shards]
        assert max(load) - min(load) <= 2 * 512 * 704
        for sh in shards:
            bs = R.make_batches(sh, 8)
            assert sorted(it.order for b in bs for it in b) == sorted(it.order for it in sh)
            assert all(len({(it.height, it.width) for it in b}) == 1 and len(b) <= 8 for b in bs)


def test_noise_slices_equal_the_sequential_stream():
    """Sharding / batching must not change any item's noise: each item's latent noise equals
    what the reference's sequential per-variant draws from torch.manual_seed(SEED) give."""
    items = [R.WorkItem(k, k, "", "", 0, "", "", 512, [512, 704, 768][k % 3]) for k in range(9)]
    items[3].skip = True
    for dtype in (torch.float16, torch.float32):
        g = torch.manual_seed(1)
        ref = {it.order: torch.randn((1, 4, 64, it.width // 8), generator=g, dtype=dtype) for it in items if not it.skip}
        for world in (1, 2, 4):
            for sh in R.shard_items(items, world):
                got = R.noise_for_items(items, sh, 1, dtype)
                assert set(got) == {it.order for it in sh}
                for k, v in got.items():
                    assert torch.equal(v, ref[k])


def test_tokenizers():
    from saspa_aug_amd.tokenizer import HashTokenizer
    t = HashTokenizer()
    a = t("An airplane, a painting of monet")
    assert a.shape == (1, 77) and a[0, 0] == 49406 and a[0, -1] == 49407
    assert np.array_equal(a, t("an  AIRPLANE , a painting of monet"))
    assert (t("x " * 200)[0] != 49407).sum() == 76          # truncated to 75 words + BOS


def test_scheduler_known_answers():
    """DDIM 'leading' timesteps with steps_offset=1 and alpha-bar values."""
    from saspa_aug_amd.scheduler import DDIMScheduler
    s = DDIMScheduler()
    assert s.set_timesteps(50)[:3].tolist() == [981, 961, 941] and s.timesteps[-1] == 1
    assert s.set_timesteps(30)[:3].tolist() == [958, 925, 892]
    assert s.set_timesteps(10).tolist() == [901, 801, 701, 601, 501, 401, 301, 201, 101, 1]
    assert abs(float(s.alphas_cumprod[0]) - 0.99915) < 1e-6 and abs(float(s.alphas_cumprod[999]) - 0.0046602) < 1e-6
    sa_t, s1_t, sa_p, s1_p = s.step_coefficients(1)          # last step: alpha_prev = alphas_cumprod[0]
    assert abs(sa_p ** 2 - 0.99915) < 1e-6 and abs(sa_t ** 2 + s1_t ** 2 - 1) < 1e-6
    from oracle.pipeline import DDIM
    o = DDIM()
    o.set_timesteps(10)
    for t in s.timesteps:
        a_t, a_p = o.coefficients(t)
        c = s.step_coefficients(t)
        assert c == (float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5))


def test_pipeline_refuses_cpu():
    from saspa_aug_amd.config import tiny
    from saspa_aug_amd.pipeline import StableDiffusionControlNetPipeline
    from saspa_aug_amd import weights as W
    cf = tiny()
    pipe = StableDiffusionControlNetPipeline(W.synth_family(cf, 0), cf)
    with pytest.raises(RuntimeError):
        pipe.to("cpu", torch.float32)
    with pytest.raises(RuntimeError):
        pipe(prompt="x", image=Image.new("RGB", (64, 64)))
    with pytest.raises(NotImplementedError):
        R.init_pipeline("sd_xl", "canny", 0)                # SDXL base + refiner is a baseline branch
    from saspa_aug_amd.config import tiny_xl
    from saspa_aug_amd.pipeline import StableDiffusionXLControlNetPipeline
    cx = tiny_xl()
    xl = R.init_pipeline("sd_xl-turbo", "canny", 0, cfgs=cx, state_dicts=W.synth_family(cx, 0))
    assert isinstance(xl, StableDiffusionXLControlNetPipeline) and xl.scheduler.config["timestep_spacing"] == "trailing"
    with pytest.raises(RuntimeError):
        xl.to("cpu", torch.float16)
    with pytest.raises(NotImplementedError):
        R.init_pipeline("blip_diffusion", "canny", 1)       # SDEdit is a baseline branch


def test_png_writer_processes_write_what_pillow_reads(tmp_path):
    """run_aug's PNG writers are png_worker.py child processes fed through pipes (Pillow's encoder holds the GIL): the
    files must decode to exactly the submitted pixels, for RGB and single-channel arrays and paths with spaces."""
    rs = np.random.RandomState(0)
    imgs = [rs.randint(0, 256, (40 + 8 * i, 64, 3)).astype(np.uint8) for i in range(6)] + [rs.randint(0, 256, (32, 48)).astype(np.uint8)]
    for procs in ("3", "0"):
        os.environ["SASPA_PNG_PROCS"] = procs
        try:
            w = R._PngWriters(3)
            paths = [tmp_path / f"w{procs} img {i}_prompt_a photo, of x_{i}.png" for i in range(len(imgs))]
            for im, pth in zip(imgs, paths):
                w.submit(im, pth)
            w.close()
        finally:
            del os.environ["SASPA_PNG_PROCS"]
        for im, pth in zip(imgs, paths):
            assert np.array_equal(np.asarray(Image.open(pth)), im)


def test_chunk_major_packing_is_the_documented_permutation():
    from saspa_aug_amd import weights as W
    co, ci, taps = 5, 128, 9
    w = torch.arange(co * ci * taps, dtype=torch.float32).reshape(co, ci, 3, 3)
    tap_major = W.pack_conv(w)                                   # K = tap * C + c
    for dtype, bk in ((torch.bfloat16, 64), (torch.float32, 32)):
        assert W.ktile(dtype) == bk and W.chunk_major_ok(3, 3, ci, 0, dtype) and not W.chunk_major_ok(1, 1, ci, 0, dtype)
        cm = W.to_chunk_major(tap_major, taps, dtype)            # K = (chunk * taps + tap) * bk + c_in_chunk
        for n in (0, 4):
            for chunk in range(ci // bk):
                for tap in (0, 4, 8):
                    for c in (0, bk - 1):
                        assert cm[n, (chunk * taps + tap) * bk + c] == tap_major[n, tap * ci + chunk * bk + c] == w[n, chunk * bk + c, tap // 3, tap % 3]
    assert not W.chunk_major_ok(3, 3, 96, 0, torch.bfloat16) and W.chunk_major_ok(3, 3, 96, 32, torch.float32)

"""End-to-end run of the generation loop (saspa_aug_amd.run_aug.main) on the device with reduced-width synthetic
weights and an on-disk synthetic dataset: both built model families, output tree, side files and the JSON contract."""
import json
from pathlib import Path

import pytest
import torch

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import config as CFG
from saspa_aug_amd import run_aug as R
from saspa_aug_amd import weights as W
from saspa_aug_amd.pipeline import (BlipDiffusionControlNetPipeline, StableDiffusionControlNetPipeline,
                                    StableDiffusionXLControlNetPipeline)

pytestmark = pytest.mark.gpu


def _settings(tmp_path, base_model, **kw):
    prompts = tmp_path / "prompts.txt"
    prompts.write_text("".join(f"an airplane in scene {k}.\n" for k in range(6)))
    return R.Settings(DATASET="synthetic", BASE_MODEL=base_model, RESOLUTION=64, NUM_INFERENCE_STEPS=3, NUM_PER_IMAGE=2, SEED=1,
                      USE_ARTISTIC_PROMPTS=(base_model == "sd_v1.5"), SEMANTIC_FILTERING=0, MODEL_CONFIDENCE_BASED_FILTERING=0,
                      PROMPTS_FILE=str(prompts), BATCH_SIZE=4,
                      DATASET_KWARGS=dict(root_path=str(tmp_path / "ds" / "data"), n_images=5, sizes=((64, 64), (64, 128))), **kw)


@pytest.mark.parametrize("base_model", ["sd_v1.5", "blip_diffusion"])
def test_run_aug_end_to_end(dev, tmp_path, base_model):
    cfgs = CFG.tiny()
    fam = W.synth_family(cfgs, seed=3)
    cls = BlipDiffusionControlNetPipeline if base_model == "blip_diffusion" else StableDiffusionControlNetPipeline
    pipe = cls(fam, cfgs).to("cuda:0", torch.float16)
    s = _settings(tmp_path, base_model)
    res = R.main(s, pipe=pipe)
    out = Path(res["output_folder"])
    assert f"aug_data/controlnet/{base_model}/canny/" in str(out)
    assert (res["status"] == 1).all() and len(res["items"]) == 10
    pngs = sorted(p.name for p in out.glob("*.png"))
    gen = [n for n in pngs if "_prompt_" in n]
    assert len(gen) == 10 and len([n for n in pngs if n.endswith("_source.png")]) == 5
    assert len([n for n in pngs if n.endswith("_control.png")]) == 5
    if base_model == "blip_diffusion":
        assert len([n for n in pngs if "_subject_" in n]) == 10
        assert all(it.subject_path for it in res["items"])
    body = json.load(open(res["json_path"]))
    assert len(body) == 5 and all(len(v) == 2 for v in body.values())
    assert all("_subject_" not in p and "_source" not in p and "_control" not in p for v in body.values() for p in v)
    # a second run finds every output on disk and generates nothing
    res2 = R.main(s, pipe=pipe)
    assert all(it.skip for it in res2["items"]) and (res2["status"] == 0).all()


def test_run_aug_end_to_end_sdxl_turbo(dev, tmp_path):
    """The sd_xl-turbo settings of the reference (run_aug/run_aug.py:567-571: guidance 0, 2 steps, no negative prompt)
    through the same loop: output tree under controlnet/sd_xl-turbo/canny, JSON contract unchanged."""
    cfgs = CFG.tiny_xl()
    fam = W.synth_family(cfgs, seed=3)
    pipe = R.init_pipeline("sd_xl-turbo", "canny", 0, cfgs=cfgs, state_dicts=fam).to("cuda:0", torch.float16)
    assert isinstance(pipe, StableDiffusionXLControlNetPipeline) and pipe.vae.dtype == torch.float32
    s = _settings(tmp_path, "sd_xl-turbo", GUIDANCE_SCALE=0)
    s.NUM_INFERENCE_STEPS = 2
    res = R.main(s, pipe=pipe)
    out = Path(res["output_folder"])
    assert "aug_data/controlnet/sd_xl-turbo/canny/" in str(out)
    assert (res["status"] == 1).all() and len(res["items"]) == 10
    body = json.load(open(res["json_path"]))
    assert len(body) == 5 and all(len(v) == 2 for v in body.values())


def test_run_aug_end_to_end_hed_control(dev, tmp_path):
    """CONTROLNET = "hed" (run_aug/run_aug.py:311-312, :438-439): the control images come from the HED annotator (batched, on
    the device), the tree moves to controlnet/sd_v1.5/hed/, everything else is unchanged."""
    import numpy as np
    from PIL import Image
    cfgs = CFG.tiny()
    pipe = R.init_pipeline("sd_v1.5", "hed", 0, cfgs=cfgs, state_dicts=W.synth_family(cfgs, seed=3)).to("cuda:0", torch.float16)
    assert isinstance(pipe, StableDiffusionControlNetPipeline)
    s = _settings(tmp_path, "sd_v1.5", CONTROLNET="hed")
    s.RESOLUTION, s.NUM_PER_IMAGE, s.NUM_INFERENCE_STEPS = 512, 1, 2
    s.DATASET_KWARGS = dict(root_path=str(tmp_path / "ds" / "data"), n_images=3, sizes=((512, 512), (512, 576)))
    res = R.main(s, pipe=pipe)
    out = Path(res["output_folder"])
    assert "aug_data/controlnet/sd_v1.5/hed/" in str(out)
    assert (res["status"] == 1).all() and len(res["items"]) == 3
    ctrl = sorted(out.glob("*_control.png"))
    assert len(ctrl) == 3
    c = np.asarray(Image.open(ctrl[0]))
    # a grey-level soft edge map with three equal channels, not Canny's {0, 255}
    assert c.ndim == 3 and np.array_equal(c[..., 0], c[..., 1]) and len(np.unique(c)) > 8


def test_run_aug_main_with_filter_models(dev, tmp_path):
    """main(..., filter_models=(sem, conf)) with the filter flags set: the stage runs on s.DEVICE (it used to hand the
    batches over on the CPU when the models were passed in) and the JSON carries the filtered name."""
    from saspa_aug_amd import filters
    from saspa_aug_amd.tokenizer import HashTokenizer
    cfgs = CFG.tiny()
    pipe = StableDiffusionControlNetPipeline(W.synth_family(cfgs, seed=3), cfgs).to("cuda:0", torch.float16)
    s = _settings(tmp_path, "sd_v1.5")
    s.SEMANTIC_FILTERING, s.MODEL_CONFIDENCE_BASED_FILTERING = 1, 1
    ds = R.dataset_utils.DS_UTILS_DICT[s.DATASET](**s.DATASET_KWARGS)
    cf = CFG.tiny_filters(num_classes=6)
    sem = filters.SemanticFilter(W.synth_state_dict("clip_rn50", cf["clip_rn50"], 31), cf["clip_rn50"], dev, ds.get_basic_prompt(),
                                 HashTokenizer(cf["clip_rn50"]["vocab"], pad_id=0))
    conf = filters.ConfidenceFilter(W.synth_state_dict("cal", cf["cal"], 32), cf["cal"], dev, top_k=3)
    res = R.main(s, ds_utils=ds, pipe=pipe, filter_models=(sem, conf))
    assert (res["status"] == 1).all()
    name = Path(res["json_path"]).name
    assert "semantic_filtering" in name and "model_confidence_based_filtering_top_10_classes" in name
    body = json.load(open(res["json_path"]))
    assert len(body) == 5 and all(len(v) <= 2 for v in body.values())
    assert all(Path(p).exists() for v in body.values() for p in v)

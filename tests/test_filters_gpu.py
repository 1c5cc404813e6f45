"""Filter stage on the MI355X: the new kernels (pooling, ReLU epilogues, BAP tail, windows beyond 31 taps),
the device pre-processing against PIL, CLIP-RN50 and WSDAN_CAL launch graphs against the oracle -- and WSDAN_CAL against
logits produced by the REFERENCE's own module (tests/golden/reference_filter_golden.json) -- and the decisions / JSON."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from PIL import Image

import saspa_aug_amd  # noqa: F401
from oracle import filter_models as FM
from saspa_aug_amd import config as CFG
from saspa_aug_amd import dataset_utils as DU
from saspa_aug_amd import filters, ops, utils
from saspa_aug_amd import weights as W
from saspa_aug_amd.synthetic import synthetic_image
from saspa_aug_amd.tokenizer import HashTokenizer
from tests.util import from_nhwc, to_nhwc

pytestmark = pytest.mark.gpu
G = json.load(open(Path(__file__).parent / "golden" / "reference_filter_golden.json"))


def _rel(got, ref):
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pool2d(dev, dtype):
    x = torch.randn(3, 16, 23, 19, generator=torch.Generator().manual_seed(0))
    xd = to_nhwc(x, dtype, dev)
    xq = xd.float().cpu().permute(0, 3, 1, 2)
    for k, s, p, mode in ((3, 2, 1, "max"), (2, 2, 0, "avg"), (7, 7, 0, "avg"), (2, 1, 0, "max")):
        got = from_nhwc(ops.pool2d(xd, k, s, p, mode=mode))
        ref = F.max_pool2d(xq, k, s, p) if mode == "max" else F.avg_pool2d(xq, k, s)
        assert got.shape == ref.shape
        assert (got - ref).abs().max() < (1e-6 if dtype == torch.float32 else 2e-2), (k, s, p, mode)


def test_signsqrt_l2norm(dev):
    x = torch.randn(5, 4096, generator=torch.Generator().manual_seed(1))
    x[0, :7] = 0.0
    got = ops.signsqrt_l2norm(x.to(dev), 1e-6, 100.0).cpu()
    ref = F.normalize(torch.sign(x) * torch.sqrt(x.abs() + 1e-6), dim=-1) * 100.0
    assert (got - ref).abs().max() < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["relu", "add_relu"])
def test_relu_epilogues(dev, dtype, act):
    g = torch.Generator().manual_seed(2)
    code = ops.ACT_RELU if act == "relu" else ops.ACT_ADD_RELU
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    # linear (tiled), long-K linear at small M (split-K reduce kernel), 3x3 conv, wide-kernel conv (bf16, many rows)
    for (m, n, k) in ((300, 96, 64), (64, 160, 4096)):
        x, w, b, r = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g), torch.randn(m, n, generator=g)
        xd, wd, rd = x.to(dev, dtype), w.to(dev, dtype), r.to(dev, dtype)
        got = ops.linear(xd, wd, b.to(dev), residual=rd, act=code).float().cpu()[:, :n]
        y = xd.float().cpu() @ wd.float().cpu().t() + b
        ref = (F.relu(y) + rd.float().cpu()) if act == "relu" else F.relu(y + rd.float().cpu())
        assert (got - ref).abs().max() < tol * max(1.0, ref.abs().max().item()), (m, n, k)
    for (bsz, c, hw, n) in ((2, 64, 12, 64), (4, 128, 64, 320)):
        x, w, b = torch.randn(bsz, c, hw, hw, generator=g), torch.randn(n, c, 3, 3, generator=g) / (9 * c) ** 0.5, torch.randn(n, generator=g)
        r = torch.randn(bsz, n, hw, hw, generator=g)
        xd, rd = to_nhwc(x, dtype, dev), to_nhwc(r, dtype, dev)
        wq = w.to(dtype).float()
        got = from_nhwc(ops.conv(xd, W.pack_conv(wq).to(dev, dtype), b.to(dev), kh=3, kw=3, pad=1, residual=rd, act=code), n)
        y = F.conv2d(xd.float().cpu().permute(0, 3, 1, 2), wq, b, padding=1)
        rq = rd.float().cpu().permute(0, 3, 1, 2)
        ref = (F.relu(y) + rq) if act == "relu" else F.relu(y + rq)
        assert (got - ref).abs().max() < tol * max(1.0, ref.abs().max().item()), (bsz, c, hw, n)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_7x7_stride2_stem_and_1x1_stride2(dev, dtype):
    """49 taps (> the 31-tap bitmask of the fast loaders) on 3 -> 8 padded channels: the generic per-lane loader."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 40, 56, generator=g)
    w = torch.randn(16, 3, 7, 7, generator=g) / 12.0
    xd = to_nhwc(x, dtype, dev, cpad=8)
    wq = w.to(dtype).float()
    got = from_nhwc(ops.conv(xd, W.pack_conv(wq).to(dev, dtype), None, kh=7, kw=7, stride=2, pad=3, act=ops.ACT_RELU), 16)
    ref = F.relu(F.conv2d(xd.float().cpu()[..., :3].permute(0, 3, 1, 2), wq, stride=2, padding=3))
    assert got.shape == ref.shape and (got - ref).abs().max() < (1e-4 if dtype == torch.float32 else 3e-2)
    x = torch.randn(2, 64, 14, 14, generator=g)
    w = torch.randn(128, 64, 1, 1, generator=g) / 8.0
    xd = to_nhwc(x, dtype, dev)
    wq = w.to(dtype).float()
    got = from_nhwc(ops.conv(xd, W.pack_conv(wq).to(dev, dtype), None, kh=1, kw=1, stride=2, pad=0), 128)
    ref = F.conv2d(xd.float().cpu().permute(0, 3, 1, 2), wq, stride=2)
    assert got.shape == ref.shape and (got - ref).abs().max() < (1e-4 if dtype == torch.float32 else 3e-2)


def test_preprocessing_equals_pil(dev):
    for (h, w) in ((512, 512), (512, 704), (640, 512), (100, 333)):
        img = synthetic_image(h, w, 3)
        d = torch.from_numpy(img)[None].to(dev)
        got = from_nhwc(filters.rn50_preprocess(d, torch.float32, 224), 3)[0]
        assert (got - FM.rn50_preprocess(img)).abs().max() < 1e-5, (h, w)
        got = from_nhwc(filters.cal_preprocess(d, torch.float32, 224), 3)[0]
        assert (got - FM.cal_preprocess(img)).abs().max() < 1e-5, (h, w)


def _ids(vocab, n=7):
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(1, vocab - 2, (n, 77), generator=g)
    ids[:, 0] = vocab - 2
    for r in range(n):
        e = 4 + 3 * r
        ids[r, e] = vocab - 1
        ids[r, e + 1:] = 0
    return ids


@pytest.mark.parametrize("full", [False, True])
def test_clip_rn50_vs_oracle(dev, full):
    cfg = CFG.CLIP_RN50 if full else CFG.tiny_filters()["clip_rn50"]
    sd = W.synth_state_dict("clip_rn50", cfg, 7)
    s = cfg["image_size"]
    x = torch.randn(2, 3, s, s, generator=torch.Generator().manual_seed(5))
    ref = FM.clip_rn50_visual(sd, cfg, x)
    vis = filters.ClipRN50Visual(sd, cfg, dev, torch.float32)
    got = vis.forward(to_nhwc(x, torch.float32, dev, cpad=8)).float().cpu()[:, :cfg["embed_dim"]]
    assert _rel(got, ref) < 2e-4, _rel(got, ref)
    if not full:
        got16 = filters.ClipRN50Visual(sd, cfg, dev, torch.bfloat16).forward(to_nhwc(x, torch.bfloat16, dev, cpad=8)).float().cpu()[:, :cfg["embed_dim"]]
        assert _rel(got16, ref) < 8e-2, _rel(got16, ref)


@pytest.mark.parametrize("tag", ["tiny", "resnet50", "resnet101"])
def test_wsdan_cal_vs_oracle_and_reference_golden(dev, tag):
    if tag == "tiny":
        cfg, seed, xseed, shape = CFG.tiny_filters()["cal"], 9, 6, (3, 3, 64, 64)
    else:
        g = G[tag]
        cfg, seed, xseed, shape = dict(g["cfg"], layers=tuple(g["cfg"]["layers"])), g["weight_seed"], g["input_seed"], tuple(g["input_shape"])
    sd = W.synth_state_dict("cal", cfg, seed)
    x = torch.randn(shape, generator=torch.Generator().manual_seed(xseed))
    with torch.no_grad():
        ref = FM.wsdan_cal_logits(sd, cfg, x)
    got = filters.WSDANCAL(sd, cfg, dev, torch.float32).forward(to_nhwc(x, torch.float32, dev, cpad=8)).float().cpu()
    assert _rel(got, ref) < 3e-4, _rel(got, ref)
    if tag != "tiny":                              # the reference's own WSDAN_CAL.forward on the same weights / input
        gold = torch.tensor(G[tag]["logits"]).float()
        assert _rel(got, gold) < 3e-4, _rel(got, gold)
        assert torch.equal(got.argsort(-1, descending=True)[:, :10], gold.argsort(-1, descending=True)[:, :10])


def test_filter_decisions_and_json_end_to_end(dev, tmp_path):
    """Real (reduced-width) filter models on the device through create_json: the lists in the JSON are exactly the images
    the ORACLE models pass (PIL pre-processing, torch-CPU networks), confidence filter first, semantic second."""
    cf = CFG.tiny_filters(num_classes=6)
    root = tmp_path / "ds/data"
    ds = DU.SyntheticUtils(root_path=str(root), n_images=6, sizes=((64, 64),), print_func=lambda *a, **k: None)
    folder = root / "aug_data/controlnet/sd_v1.5/canny/run_seed_1/images"
    folder.mkdir(parents=True)
    files = {}
    for k, p in enumerate(ds.original_images_paths):
        stem = Path(p).stem
        for v in range(3):
            img = synthetic_image(64 if v < 2 else 96, 64 if v < 2 else 128, 100 + 10 * k + v)
            name = f"{stem}_prompt_An airplane, oil painting_{v}.png"
            Image.fromarray(img).save(folder / name)
            files[str(folder / name)] = (Path(p).name, img)
        Image.fromarray(synthetic_image(64, 64, k)).save(folder / f"{stem}_source.png")
    sd_c, sd_w = W.synth_state_dict("clip_rn50", cf["clip_rn50"], 31), W.synth_state_dict("cal", cf["cal"], 32)
    tok = HashTokenizer(cf["clip_rn50"]["vocab"], pad_id=0)
    sem = filters.SemanticFilter(sd_c, cf["clip_rn50"], dev, ds.get_basic_prompt(), tok)
    conf = filters.ConfidenceFilter(sd_w, cf["cal"], dev, top_k=3)
    # oracle decisions per augmented image
    labels = ds.get_image_path_to_class_id_dict()
    by_name = {Path(p).name: labels[p] for p in ds.original_images_paths}
    ids = torch.from_numpy(np.concatenate([tok(pr) for pr in sem.prompts]))
    ok_c, ok_s = {}, {}
    for path, (orig, img) in files.items():
        with torch.no_grad():
            lg_c = FM.wsdan_cal_logits(sd_w, cf["cal"], FM.cal_preprocess(img, (64, 64))[None])[0]
            lg_s = FM.clip_selector_logits(sd_c, cf["clip_rn50"], FM.rn50_preprocess(img, 64)[None], ids)[0]
        ok_c[path] = FM.confidence_pass(lg_c[None], by_name[orig], 3)
        ok_s[path] = bool(FM.semantic_pass(lg_s[None])[0])

    def expected(use_s, use_c):
        want = {Path(p).name: [] for p in ds.original_images_paths}
        for path, (orig, _) in files.items():
            if (ok_c[path] or not use_c) and (ok_s[path] or not use_s):
                want[orig].append(path)
        return {k: sorted(v) for k, v in want.items()}

    for use_s, use_c in ((0, 1), (1, 0), (1, 1)):
        jp = utils.create_json_of_image_name_to_augmented_images_paths(
            ds, str(folder), semantic_filtering=use_s, model_confidence_based_filtering=use_c, conf_top_k=3, init_log=False,
            original_images_paths=ds.original_images_paths, min_files=1, filter_models=(sem, conf), device=dev)
        assert ("model_confidence_based_filtering_top_3_classes" in Path(jp).name) == bool(use_c)
        assert ("semantic_filtering" in Path(jp).name) == bool(use_s)
        got = {k: sorted(v) for k, v in json.load(open(jp)).items()}
        assert got == expected(use_s, use_c), (use_s, use_c)
    n_kept = sum(ok_c.values())
    assert 0 < n_kept < len(files), "the synthetic classifier should drop some images and keep some"

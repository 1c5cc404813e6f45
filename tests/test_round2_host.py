"""Host-side fixes of round 2: legacy VAE attention key names, EXIF-consistent planning, CLIP BPE
pre-tokenisation against transformers' CLIPTokenizer on a synthetic vocabulary."""
import json

import numpy as np
import pytest
import torch
from PIL import Image

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import config as CFG
from saspa_aug_amd import run_aug as R
from saspa_aug_amd import weights as W
from saspa_aug_amd.tokenizer import CLIPBPETokenizer, _bytes_to_unicode


def test_legacy_vae_attention_keys_are_remapped(tmp_path):
    """runwayml/stable-diffusion-v1-5's vae/ checkpoint predates diffusers 0.7: AttentionBlock names query/key/value/
    proj_attn (diffusers renames them at load, `_convert_deprecated_attention_blocks`); conv-shaped [C,C,1,1] weights of the
    even older form are squeezed."""
    from safetensors.torch import save_file
    cfg = CFG.tiny()["vae"]
    sd = W.synth_state_dict("vae", cfg, 0)
    a = "decoder.mid_block.attentions.0"
    legacy = {}
    names = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    for k, v in sd.items():
        for new, old in names.items():
            if k.startswith(f"{a}.{new}."):
                k = k.replace(f"{a}.{new}.", f"{a}.{old}.")
                if k.endswith("weight") and old != "value":
                    v = v[:, :, None, None]            # mix conv-shaped and linear-shaped legacy weights
        legacy[k] = v.contiguous()
    assert f"{a}.query.weight" in legacy and f"{a}.to_q.weight" not in legacy
    path = tmp_path / "diffusion_pytorch_model.safetensors"
    save_file(legacy, str(path))
    got = W.load_safetensors(str(path))
    assert set(got) == set(sd)
    for k in sd:
        assert got[k].shape == sd[k].shape and torch.equal(got[k], sd[k]), k
    # current names pass through untouched; a checkpoint holding both forms is refused
    assert set(W.remap_legacy_attention_keys(sd)) == set(sd)
    both = dict(sd)
    both[f"{a}.query.weight"] = sd[f"{a}.to_q.weight"]
    with pytest.raises(KeyError):
        W.remap_legacy_attention_keys(both)


def test_plan_uses_the_exif_transposed_size(tmp_path):
    """A JPEG stored 128x64 with EXIF orientation 6 loads as 64x128 (load_source / diffusers.load_image transpose it): the
    planned bucket and noise shape must be those of the loaded image."""
    img = Image.fromarray(np.random.RandomState(0).randint(0, 255, (64, 128, 3), np.uint8))
    exif = Image.Exif()
    exif[0x0112] = 6
    p = tmp_path / "rot.jpg"
    img.save(p, exif=exif)
    s = R.Settings(DATASET="synthetic", NUM_PER_IMAGE=1, RESOLUTION=64, USE_ARTISTIC_PROMPTS=False, PROMPT_WITH_SUB_CLASS=False)
    np.random.seed(1)
    items = R.plan_work(s, [str(p)], ["a photo of an airplane"], str(tmp_path / "out"), {})
    loaded = R.load_source(str(p), 64)
    assert loaded.shape[:2] == (128, 64)
    assert (items[0].height, items[0].width) == loaded.shape[:2]


def _png_with_text_orientation(path, kind):
    """A 128x64 PNG that carries EXIF orientation 6 WITHOUT an eXIf chunk: as ImageMagick's `Raw profile type exif` text chunk
    (hex dump of the EXIF blob) or as an XMP packet's tiff:Orientation (iTXt) -- both are sources Pillow's getexif() reads."""
    from PIL import PngImagePlugin
    img = Image.fromarray(np.random.RandomState(0).randint(0, 255, (64, 128, 3), np.uint8))
    info = PngImagePlugin.PngInfo()
    if kind == "raw_profile":
        exif = Image.Exif()
        exif[0x0112] = 6
        raw = exif.tobytes()
        blob = raw if raw.startswith(b"Exif\x00\x00") else b"Exif\x00\x00" + raw
        hexed = blob.hex()
        body = "\nexif\n%8d\n" % len(blob) + "\n".join(hexed[i:i + 72] for i in range(0, len(hexed), 72)) + "\n"
        info.add_text("Raw profile type exif", body, zip=(kind == "raw_profile"))
    else:
        xmp = ('<?xpacket begin="" id="W5M0MpCehiHzreSzNTczkc9d"?><x:xmpmeta xmlns:x="adobe:ns:meta/"><rdf:RDF '
               'xmlns:rdf="http://www.w3.org/1999/02/22-rdf-syntax-ns#"><rdf:Description rdf:about="" '
               'xmlns:tiff="http://ns.adobe.com/tiff/1.0/" tiff:Orientation="6"/></rdf:RDF></x:xmpmeta><?xpacket end="w"?>')
        info.add_itxt("XML:com.adobe.xmp", xmp)
    img.save(path, pnginfo=info)


@pytest.mark.parametrize("kind", ["raw_profile", "xmp"])
def test_plan_follows_pillow_for_png_orientation_outside_exif_chunks(tmp_path, kind):
    """`_png_orientation` used to read the eXIf chunk only, while `load_raw` transposes by whatever
    `Image.getexif()` finds (also a `Raw profile type exif` text chunk and XMP tiff:Orientation).  Whatever Pillow decides for
    such a PNG -- transposed or not, that is version-dependent -- the planned (h, w) must be the loaded image's."""
    p = tmp_path / f"rot_{kind}.png"
    _png_with_text_orientation(p, kind)
    with Image.open(p) as im:
        pillow_says = im.getexif().get(0x0112, 1)
    assert R._png_orientation(str(p)) == pillow_says
    s = R.Settings(DATASET="synthetic", NUM_PER_IMAGE=1, RESOLUTION=64, USE_ARTISTIC_PROMPTS=False, PROMPT_WITH_SUB_CLASS=False)
    np.random.seed(1)
    items = R.plan_work(s, [str(p)], ["a photo of an airplane"], str(tmp_path / "out"), {})
    loaded = R.load_source(str(p), 64)
    assert (items[0].height, items[0].width) == loaded.shape[:2]
    if pillow_says == 6:
        assert loaded.shape[:2] == (128, 64)


def test_png_orientation_fast_path_never_opens_pillow_exif(tmp_path, monkeypatch):
    """A PNG with neither an eXIf nor a text chunk stays on the header walk (no decode: the 69 s -> 0.8 s of the configs[3] plan)."""
    p = tmp_path / "plain.png"
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(p)
    monkeypatch.setattr(Image.Image, "getexif", lambda self: (_ for _ in ()).throw(AssertionError("getexif called")))
    assert R._png_orientation(str(p)) == 1


def _synthetic_clip_vocab(tmp_path):
    b2u = _bytes_to_unicode()
    chars = [b2u[b] for b in range(256)]
    vocab = chars + [c + "</w>" for c in chars]
    merges = [("a", "i"), ("p", "l"), ("n", "e</w>"), ("ai", "r"), ("air", "pl"), ("airpl", "a"), ("airpla", "ne</w>"),
              ("é", "r"), ("1", "2"), ("o", "f</w>"), ("t", "h"), ("th", "e</w>"), ("'", "s</w>"), ("Ã", "©")]
    for a, b in merges:
        vocab.append(a + b)
    vocab += ["<|startoftext|>", "<|endoftext|>"]
    enc = {t: i for i, t in enumerate(dict.fromkeys(vocab))}
    (tmp_path / "vocab.json").write_text(json.dumps(enc), encoding="utf-8")
    (tmp_path / "merges.txt").write_text("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n", encoding="utf-8")
    return enc


def test_clip_bpe_matches_transformers_on_a_synthetic_vocabulary(tmp_path):
    """Same vocab.json / merges.txt through transformers' (slow) CLIPTokenizer and through CLIPBPETokenizer: identical ids,
    including accented letters (\\p{L}), digits one by one (\\p{N}), apostrophe suffixes, NFC composition, punctuation runs."""
    transformers = pytest.importorskip("transformers")
    _synthetic_clip_vocab(tmp_path)
    try:
        ref = transformers.CLIPTokenizer(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt"))
    except Exception as e:                                      # pragma: no cover - API drift of the installed version
        pytest.skip(f"CLIPTokenizer could not be built from files: {e}")
    ours = CLIPBPETokenizer(str(tmp_path))
    prompts = ["a photo of the airplane", "An Airplane's  wing,  12 of 3456!!", "Aérospatiale ATR-72 -- école №5", "cafe\u0301 &amp; bar",
               "", "x" * 300, "Boeing 707-320 (the airplane), über-große Straße"]
    for ptxt in prompts:
        want = ref(ptxt, padding="max_length", max_length=77, truncation=True)["input_ids"]
        got = ours(ptxt)[0].tolist()
        assert got == want, (ptxt, got[:20], want[:20])


def test_fp8_weight_quantisation_round_trip():
    """weights.quantize_fp8: per-output-channel scale = row amax / 448, OCP e4m3 bytes; dequantised within half an ulp of
    3 mantissa bits; zero rows keep scale 1."""
    import torch
    from saspa_aug_amd import weights as W
    w = torch.randn(64, 256, generator=torch.Generator().manual_seed(0)) * torch.logspace(-3, 1, 64)[:, None]
    w[5] = 0
    q, s = W.quantize_fp8(w)
    assert q.dtype == torch.uint8 and q.shape == w.shape and s.shape == (64,)
    assert torch.allclose(s[torch.arange(64) != 5], w.abs().amax(1)[torch.arange(64) != 5] / 448.0) and s[5] == 1.0
    d = W.dequantize_fp8(q, s)
    assert ((d - w).abs() <= 0.0625 * w.abs() + s[:, None] * 2 ** -9 + 1e-12).all()
    assert (d[5] == 0).all() and (q.view(torch.float8_e4m3fn).float().abs().amax(1)[torch.arange(64) != 5] == 448).all()


def test_pack_geglu_tile_layout():
    import torch
    from saspa_aug_amd import weights as W
    f, k = 192, 8
    w = torch.arange(2 * f * k, dtype=torch.float32).view(2 * f, k)
    b = torch.arange(2 * f, dtype=torch.float32)
    wp, bp = W.pack_geglu_tile(w, b, 128)
    # tile t = [values of features 64t .. 64t+63 | their gates]
    for t in range(3):
        assert torch.equal(bp[128 * t:128 * t + 64], b[64 * t:64 * t + 64])
        assert torch.equal(bp[128 * t + 64:128 * (t + 1)], b[f + 64 * t:f + 64 * t + 64])
    assert torch.equal(wp[64], w[f]) and W.pack_geglu_tile(w[:200], b[:200], 128) is None
    assert all(torch.equal(a, c) for a, c in zip(W.pack_geglu(w[:320], b[:320]), W.pack_geglu_tile(w[:320], b[:320], 160)))


def test_large_tensor_synthesis_is_deterministic_and_leaves_golden_kinds_alone():
    """weights._fill_big: tensors >= 1 Mi elements are drawn chunk-wise on a thread pool (per-chunk seeds), reproducibly; the
    kinds behind reference goldens (cal, clip_rn50, hed) stay on the sequential generator."""
    import torch
    from saspa_aug_amd import config as CFG
    from saspa_aug_amd import weights as W
    cfg = dict(CFG.SD15["text"])
    a, b = W.synth_state_dict("text", cfg, 3), W.synth_state_dict("text", cfg, 3)
    big = "text_model.embeddings.token_embedding.weight"
    assert a[big].numel() >= W._BIG and all(torch.equal(a[k], b[k]) for k in a)
    c = W.synth_state_dict("text", cfg, 4)
    assert not torch.equal(a[big], c[big])
    assert abs(float(a[big].std()) - 0.5) < 5e-3 and abs(float(a[big].mean())) < 1e-3
    # an exempt kind: identical to drawing everything from ONE sequential generator
    g = torch.Generator().manual_seed(21)
    hed = W.synth_state_dict("hed", CFG.HED, 21)
    first = 110.0 + 20.0 * torch.rand((1, 3, 1, 1), generator=g)
    assert torch.equal(hed["norm"], first)
    w0 = torch.randn((64, 3, 3, 3), generator=g) * (1.0 / (27 ** 0.5))
    assert torch.equal(hed["block1.convs.0.weight"], w0)
    assert hed["block5.convs.2.weight"].numel() >= W._BIG          # large, and still sequential

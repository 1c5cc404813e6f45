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

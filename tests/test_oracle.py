"""Pins on the oracle itself (CPU).  The reference holds no golden vectors for this path and
diffusers / cv2 are not installable here (oracle header: PARITY UNPINNED), so the oracle is
checked against what IS available:
  * an independent implementation of the CLIP text tower (transformers.CLIPTextModel, random
    init, same state dict) -- tolerance 1e-4 abs on O(1) activations (fp32 op-order only);
  * hand-computed Canny cases and structural properties (edges subset of NMS survivors,
    hysteresis idempotence, threshold monotonicity, replicate border);
  * a dense-formula restatement of the DDIM update."""
import numpy as np
import pytest
import torch

from oracle import canny as OC
from oracle import pipeline as OP
from oracle import sd_models as OM

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import weights as W


def test_clip_text_matches_transformers():
    tr = pytest.importorskip("transformers")
    cfg = dict(vocab=1000, width=64, layers=3, heads=4, mlp=256, max_pos=77)
    hf_cfg = tr.CLIPTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=256, num_hidden_layers=3,
                               num_attention_heads=4, max_position_embeddings=77, hidden_act="quick_gelu",
                               layer_norm_eps=1e-5, bos_token_id=998, eos_token_id=999, pad_token_id=999)
    model = tr.CLIPTextModel(hf_cfg).eval()
    sd = W.synth_state_dict("text", cfg, seed=4)
    hf_sd = sd
    if not any(k.startswith("text_model.") for k in model.state_dict()):      # transformers >= 5 dropped the prefix
        hf_sd = {k[len("text_model."):]: v for k, v in sd.items()}
    missing = model.load_state_dict(hf_sd, strict=False)
    assert not [k for k in missing.missing_keys if "position_ids" not in k], missing
    assert not missing.unexpected_keys
    ids = torch.from_numpy(np.random.RandomState(0).randint(0, 998, (2, 77)))
    with torch.no_grad():
        ref = model(input_ids=ids).last_hidden_state
        got = OM.clip_text_forward(sd, cfg, ids)
    assert (ref - got).abs().max().item() < 1e-4


def test_canny_hand_cases():
    img = np.zeros((12, 12, 3), np.uint8)
    assert OC.canny(img, 120, 200).sum() == 0                     # flat image: no edges
    img[:, 6:] = 255                                              # vertical step edge
    e = OC.canny(img, 120, 200)
    # |dx| = 1020 on the two columns adjacent to the step; NMS '>' left / '>=' right keeps the left one
    assert (e[:, 5] == 255).all() and e[:, 6].sum() == 0 and e[:, :5].sum() == 0 and e[:, 7:].sum() == 0
    # only one channel carries the edge: the max-magnitude channel must be picked
    img2 = np.zeros((12, 12, 3), np.uint8)
    img2[:, 6:, 2] = 255
    assert np.array_equal(OC.canny(img2, 120, 200), e)
    # weak edge (between thresholds) survives only when connected to a strong one
    img3 = np.zeros((16, 16), np.uint8)
    img3[:, 8:] = 40                                              # |dx| = 160: weak
    assert OC.canny(img3, 120, 200).sum() == 0
    img3[:4, 8:] = 255                                            # strong on the first rows, same column
    e3 = OC.canny(img3, 120, 200)
    # the weak segment is 8-connected to the strong one through the corner at rows 3-4
    assert (e3[:3, 7] == 255).all() and (e3[6:, 7] == 255).all()


def test_canny_properties():
    from saspa_aug_amd.synthetic import synthetic_image
    img = synthetic_image(96, 128, 3)
    m = OC.canny_nms_map(img, 120, 200)
    e = OC.canny(img, 120, 200)
    assert set(np.unique(e)) <= {0, 255}
    assert ((e == 255) <= (m != 1)).all()                         # edges only where NMS kept the pixel
    assert ((m == 2) <= (e == 255)).all()                         # every strong pixel is an edge
    assert np.array_equal(OC.hysteresis(OC.hysteresis(m)), OC.hysteresis(m))       # idempotent
    assert np.array_equal(OC.canny(img, 200, 120), e)             # thresholds are swapped if reversed
    hi = OC.canny(img, 150, 250)
    assert ((hi == 255) <= (OC.canny(img, 120, 250) == 255)).all()               # lower 'low' can only add edges
    dx, dy = OC.sobel3_replicate(img)
    assert np.abs(dx).max() <= 1020 and np.abs(dy).max() <= 1020
    out3 = OC.generate_canny_array(img, 120, 200)
    assert out3.shape == img.shape and (out3[..., 0] == out3[..., 1]).all() and (out3[..., 0] == e).all()


def test_ddim_step_dense_formula():
    s = OP.DDIM()
    s.set_timesteps(50)
    g = torch.Generator().manual_seed(0)
    x, eps = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    for t in (981, 501, 1):
        a_t, a_p = s.coefficients(t)
        want = (a_p / a_t).sqrt() * x + ((1 - a_p).sqrt() - (a_p * (1 - a_t) / a_t).sqrt()) * eps
        assert torch.allclose(s.step(eps, t, x), want, atol=1e-5)


def test_timestep_embedding_layout():
    e = OM.timestep_sinusoid(torch.tensor([0.0, 7.0]), 320)
    assert e.shape == (2, 320) and torch.allclose(e[0, :160], torch.ones(160)) and torch.allclose(e[0, 160:], torch.zeros(160))
    assert abs(e[1, 0].item() - np.cos(7.0)) < 1e-6 and abs(e[1, 160].item() - np.sin(7.0)) < 1e-6


def test_param_counts_match_public_figures():
    from saspa_aug_amd import config as CFG
    n = {k: sum(torch.Size(s).numel() for _, s, _ in W.SPECS[k](CFG.SD15[k])) for k in CFG.SD15}
    assert n == dict(unet=859520964, controlnet=361279120, vae=49490199, text=123060480)
    assert CFG.SD15_UNET == OM.SD15_UNET and CFG.SD15_CONTROLNET == OM.SD15_CONTROLNET and CFG.SD15_VAE == OM.SD15_VAE

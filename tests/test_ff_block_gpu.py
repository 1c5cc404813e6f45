"""`saspa_ff_block` (round 6): LayerNorm -> GEGLU projection -> output projection -> residual of a level-0 transformer block in one
launch, against (a) a PyTorch fp32 reference with the same rounding points (LayerNorm output, the gated hidden state and the Linear
output are bf16 in both paths) and (b) the two launches it replaces (A-stationary LayerNorm + GEGLU, output projection + residual)."""
import math

import pytest
import torch
import torch.nn.functional as F

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
from saspa_aug_amd import weights as W

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _operands(f, seed):
    w1 = _rand(2 * f, 320, seed=seed, scale=1 / math.sqrt(320))
    b1 = _rand(2 * f, seed=seed + 1, scale=0.2)
    w2 = _rand(320, f, seed=seed + 2, scale=1 / math.sqrt(f))
    b2 = _rand(320, seed=seed + 3, scale=0.2)
    g = 1 + 0.1 * _rand(320, seed=seed + 4)
    be = 0.1 * _rand(320, seed=seed + 5)
    return w1, b1, w2, b2, g, be


def _reference(x, w1, b1, w2, b2, g, be, ln=True):
    """fp32 arithmetic on the bf16-rounded operands, bf16 roundings where both device paths round."""
    xf = x.float()
    n = F.layer_norm(xf, (320,), g, be, 1e-5).to(BF).float() if ln else xf
    f = w2.shape[1]
    a = n @ w1.to(BF).float().t() + b1
    hid = (a[:, :f] * F.gelu(a[:, f:])).to(BF).float()
    y = (hid @ w2.to(BF).float().t() + b2).to(BF).float()
    return (y + xf).to(BF)


@pytest.mark.parametrize("ws", ["0", "1"])
@pytest.mark.parametrize("m,f,ln", [(256, 1280, True), (1024, 1280, True), (128, 64, True), (384, 1280, False), (4096, 320, True)])
def test_ff_block_vs_reference(dev, m, f, ln, ws, monkeypatch):
    # ws = "1": the wave-specialised form (SASPA_FF_WS: 8 waves, rows / accumulators split over wave pairs); both forms compute the
    # same sums in the same order -> compared with each other bit for bit below
    monkeypatch.setenv("SASPA_FF_WS", ws)
    x = (_rand(m, 320, seed=1) * 1.5 + 0.2).to(BF)
    w1, b1, w2, b2, g, be = _operands(f, 10)
    w1p, b1p, w2f, b2p = W.pack_ff_block(w1, b1, w2, b2)
    out = ops.ff_block(x.to(dev), (g.to(dev), be.to(dev), 1e-5) if ln else None, w1p.to(dev, BF), b1p.to(dev), w2f.to(dev, BF), b2p.to(dev))
    ref = _reference(x, w1, b1, w2, b2, g, be, ln)
    got = out.cpu().float()
    err = (got - ref.float()).abs()
    scale = ref.float().abs().max().item()
    # bf16 output rounding (2^-8 relative) + the fp32 summation order + rare rounding flips of the bf16 hidden state
    assert err.max().item() <= 2.5e-2 * scale, (err.max().item(), scale)
    assert (err.pow(2).mean().sqrt() / ref.float().pow(2).mean().sqrt()).item() < 4e-3
    if ws == "1":
        monkeypatch.setenv("SASPA_FF_WS", "0")
        four = ops.ff_block(x.to(dev), (g.to(dev), be.to(dev), 1e-5) if ln else None, w1p.to(dev, BF), b1p.to(dev), w2f.to(dev, BF), b2p.to(dev))
        assert torch.equal(four, out)


def test_ff_block_vs_the_two_launches_it_replaces(dev):
    m, f = 2048, 1280
    x = (_rand(2, m // 2, 320, seed=2) * 1.2).to(BF).to(dev)
    w1, b1, w2, b2, g, be = _operands(f, 20)
    w1p, b1p, w2f, b2p = W.pack_ff_block(w1, b1, w2, b2)
    fused = ops.ff_block(x, (g.to(dev), be.to(dev), 1e-5), w1p.to(dev, BF), b1p.to(dev), w2f.to(dev, BF), b2p.to(dev))
    assert fused.shape == x.shape
    wg, bg = W.pack_geglu(w1, b1)
    n3 = ops.layernorm(x, g.to(dev), be.to(dev))
    hid = ops.linear(n3, wg.to(dev, BF), bg.to(dev), act=ops.ACT_GEGLU)
    two = ops.linear(hid, w2.to(BF).to(dev), b2.to(dev), residual=x)
    d = (fused.float() - two.float())
    rel = (d.pow(2).mean().sqrt() / two.float().pow(2).mean().sqrt()).item()
    assert rel < 3e-3, rel
    # a separate residual tensor and an output buffer with a wider pitch
    res = (_rand(m, 320, seed=3)).to(BF).to(dev)
    big = torch.zeros(m, 640, device=dev, dtype=BF)
    ops.ff_block(x.reshape(m, 320), (g.to(dev), be.to(dev), 1e-5), w1p.to(dev, BF), b1p.to(dev), w2f.to(dev, BF), b2p.to(dev), residual=res, out=big[:, :320])
    want = (fused.reshape(m, 320).float() - x.reshape(m, 320).float() + res.float())
    assert (big[:, :320].float() - want).abs().max().item() <= 3e-2 * want.abs().max().item()
    assert (big[:, 320:] == 0).all()


def test_ff_block_refuses_what_it_cannot_run(dev):
    x = torch.zeros(100, 320, device=dev, dtype=BF)                      # rows not a multiple of 128
    w1, b1, w2, b2, g, be = _operands(64, 30)
    w1p, b1p, w2f, b2p = W.pack_ff_block(w1, b1, w2, b2)
    assert not ops.ff_block_eligible(x, w1p.to(dev, BF), w2f.to(dev, BF))
    with pytest.raises(RuntimeError):
        ops.ff_block(x, None, w1p.to(dev, BF), b1p.to(dev), w2f.to(dev, BF), b2p.to(dev))
    assert ops.ff_block_eligible(torch.zeros(128, 320, device=dev, dtype=BF), w1p.to(dev, BF), w2f.to(dev, BF))

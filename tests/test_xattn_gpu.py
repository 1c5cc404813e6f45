"""saspa_xattn_block (csrc/saspa_xattn.hip): LayerNorm -> to_q -> attention over the text keys -> to_out + residual of a level-0
transformer block in one launch, against (a) a plain PyTorch fp32 reference of the same chain with the same bf16 rounding
points (LayerNorm output, Q, O) and (b) the three-launch path it replaces (A-stationary LayerNorm + to_q, flash attention,
A-stationary to_out + residual).  Stands for BasicTransformerBlock.norm2 / attn2 of diffusers behind run_aug/run_aug.py:278."""
import math

import numpy as np
import pytest
import torch

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
from saspa_aug_amd import weights as W
from saspa_aug_amd.models import ATTN_LOG2E, attention_core, project_vt

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
C, HEADS, D = 320, 8, 40


def _case(dev, nsamp, ntok, nk, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(nsamp * ntok, C, generator=g) * 1.5 + 0.3).to(BF)
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    wq = torch.randn(C, C, generator=g) / math.sqrt(C)
    wo = torch.randn(C, C, generator=g) / math.sqrt(C)
    bo = 0.2 * torch.randn(C, generator=g)
    k = (torch.randn(nsamp, nk, C, generator=g) * 2.0).to(BF)          # sharp-ish softmax rows
    v = torch.randn(nsamp, nk, C, generator=g).to(BF)
    qs = D ** -0.5 * ATTN_LOG2E
    return dict(x=x, gamma=gamma, beta=beta, wq=wq * qs, wo=wo, bo=bo, k=k, v=v, nsamp=nsamp, ntok=ntok, nk=nk)


def _reference(c):
    x = c["x"].float()
    xn = torch.nn.functional.layer_norm(x, (C,), c["gamma"], c["beta"], 1e-5).to(BF).float()
    q = (xn @ c["wq"].to(BF).float().T).to(BF).float()                  # log2-domain logits: the scale is folded into wq
    nsamp, ntok, nk = c["nsamp"], c["ntok"], c["nk"]
    qh = q.view(nsamp, ntok, HEADS, D).permute(0, 2, 1, 3)
    kh = c["k"].float().view(nsamp, nk, HEADS, D).permute(0, 2, 1, 3)
    vh = c["v"].float().view(nsamp, nk, HEADS, D).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2)
    p = torch.exp2(s - s.amax(-1, keepdim=True))
    o = (p @ vh) / p.sum(-1, keepdim=True)
    o = o.permute(0, 2, 1, 3).reshape(nsamp * ntok, C).to(BF).float()
    return (o @ c["wo"].to(BF).float().T + c["bo"] + x)


def _fused(c, dev):
    w, bias = W.pack_xattn_w(c["wq"], c["wo"], c["bo"])
    kf, vf = W.xattn_kv_fragments(c["k"].to(dev), c["v"].to(dev))
    return ops.xattn_block(c["x"].to(dev), (c["gamma"].to(dev), c["beta"].to(dev), 1e-5), w.to(dev, BF), bias.to(dev), kf, vf, c["nk"],
                           c["ntok"])


@pytest.mark.parametrize("nsamp,ntok,nk", [(2, 512, 77), (1, 256, 77), (3, 768, 64), (2, 256, 96), (2, 256, 33), (1, 256, 1)])
def test_xattn_block_vs_reference(dev, nsamp, ntok, nk):
    c = _case(dev, nsamp, ntok, nk, 7 + nk)
    got = _fused(c, dev).float().cpu()
    ref = _reference(c)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    rms = ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"xattn_block {nsamp}x{ntok} tokens, {nk} keys: max-rel {err:.3e} rms-rel {rms:.3e}")
    assert torch.isfinite(got).all()
    assert err < 2e-2 and rms < 6e-3, (err, rms)             # bf16 output rounding is 2^-9 of the value


def test_xattn_block_vs_three_launch_path(dev):
    """The launches it replaces, on the same operands: LayerNorm + to_q (A-stationary kernel where eligible, else LayerNorm +
    linear), flash attention over the 77 keys (prescaled queries), to_out + residual."""
    nsamp, ntok, nk = 2, 4096, 77
    c = _case(dev, nsamp, ntok, nk, 3)
    got = _fused(c, dev).float()
    x = c["x"].to(dev).view(nsamp, ntok, C)
    wq, wo = c["wq"].to(dev, BF), c["wo"].to(dev, BF)
    n2 = ops.layernorm(x, c["gamma"].to(dev), c["beta"].to(dev))
    q = ops.linear(n2, wq)
    kd = c["k"].to(dev)
    vt = torch.zeros((nsamp, C, 80), device=dev, dtype=BF)
    vt[:, :, :nk] = c["v"].to(dev).transpose(1, 2)
    o = attention_core(q, kd, vt, HEADS, ntok, nk, prescaled=True)
    ref = ops.linear(o, wo, c["bo"].to(dev), residual=x).float().view(-1, C)
    rms = ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"xattn_block vs the three-launch path: max-rel {err:.3e} rms-rel {rms:.3e}")
    assert err < 2e-2 and rms < 6e-3, (err, rms)


def test_xattn_block_is_deterministic_and_validates(dev):
    c = _case(dev, 2, 256, 77, 5)
    a, b = _fused(c, dev), _fused(c, dev)
    assert torch.equal(a, b)
    w, bias = W.pack_xattn_w(c["wq"], c["wo"], c["bo"])
    kf, vf = W.xattn_kv_fragments(c["k"].to(dev), c["v"].to(dev))
    with pytest.raises(RuntimeError):        # 250 rows per sample: a workgroup's rows would straddle two samples
        ops.xattn_block(c["x"].to(dev)[:500], (c["gamma"].to(dev), c["beta"].to(dev), 1e-5), w.to(dev, BF), bias.to(dev), kf, vf, 77, 250)

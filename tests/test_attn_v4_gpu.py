"""flash_attn_v4_kernel (round 6): 64 queries per wave, 32-key half steps, the softmax reference level carried in the padding of the
head dimension (d = 40 -> dims 40 .. 47: K's first pad element is 1, Q's is -level).  Against the fp32 reference of the same op
(tests/test_kernels_gpu.py: _ref_attn on the bf16-rounded, prescaled queries) and against the v3 loop it replaces at d = 40."""
import pytest
import torch

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
from tests.test_kernels_gpu import _prescaled_q, _rand, _ref_attn, assert_close, q

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return torch.device("cuda:0")


def _run(dev, qs, kk, vv, heads, d, nq, nk):
    dtype = torch.bfloat16
    bsz, c = qs.shape[0], heads * d
    ldvt = ops.round8(nk) + 8
    vt = torch.full((bsz, c, ldvt), float("nan"), device=dev, dtype=dtype)          # pad columns must be ignored
    vt[:, :, :nk] = vv.transpose(1, 2).to(dev, dtype)
    qkbuf = torch.zeros(bsz, max(nq, nk), 2 * c, device=dev, dtype=dtype)           # q and k interleaved like a fused projection output
    qkbuf[:, :nq, :c] = qs.to(dev, dtype)
    qkbuf[:, :nk, c:] = kk.to(dev, dtype)
    out = torch.zeros(bsz, nq, c, device=dev, dtype=dtype)
    ops.flash_attn(qkbuf[:, :nq, :c], qkbuf[:, :nk, c:], vt, out, heads, d, nq, nk, 123.0, False, prescaled=True)
    return out


@pytest.mark.parametrize("bsz,heads,nq,nk", [(2, 2, 1024, 1024), (1, 1, 257, 2048 + 31), (2, 8, 640, 512 + 77), (1, 2, 512, 512),
                                             (1, 3, 1500, 576), (1, 1, 64, 4096), (2, 1, 700, 1000)])
def test_v4_vs_reference_and_v3(dev, monkeypatch, bsz, heads, nq, nk):
    d, dtype = 40, torch.bfloat16
    monkeypatch.setenv("SASPA_ATTN_MODE", "4")
    c = heads * d
    qs, qeff = _prescaled_q(_rand(bsz, nq, c, seed=31), d)
    kk = q(_rand(bsz, nk, c, seed=32), dtype)
    vv = q(_rand(bsz, nk, c, seed=33), dtype)
    ref = _ref_attn(qeff, kk, vv, heads)
    monkeypatch.setenv("SASPA_ATTN_V4", "2")
    out4 = _run(dev, qs, kk, vv, heads, d, nq, nk)
    assert torch.isfinite(out4).all()
    assert_close(out4.float().cpu(), ref, dtype, what=f"flash v4 nq={nq} nk={nk}")
    monkeypatch.setenv("SASPA_ATTN_V4", "0")
    out3 = _run(dev, qs, kk, vv, heads, d, nq, nk)
    # same arithmetic up to the level (v4 rounds it to bf16): equal to bf16 rounding of the output
    assert (out4.float() - out3.float()).abs().max().item() <= 2 ** -6 * ref.abs().max().item()


@pytest.mark.parametrize("spike,shift", [(6.0, 0.0), (60.0, 0.0), (1.0, -40.0), (6.0, 25.0), (20.0, -25.0)])
def test_v4_level_moves(dev, monkeypatch, spike, shift):
    """The level only moves when some p reaches 2.0: force that at late half steps (spiked keys, incl. the very last key), with logits
    far below / above zero and a spike large enough that exp2 against the stale level overflows to inf."""
    monkeypatch.setenv("SASPA_ATTN_MODE", "4")
    monkeypatch.setenv("SASPA_ATTN_V4", "2")
    dtype = torch.bfloat16
    bsz, heads, d, n = 1, 1, 40, 640
    qq = _rand(bsz, n, d, seed=24)
    kk = _rand(bsz, n, d, seed=25)
    vv = q(_rand(bsz, n, d, seed=26), dtype)
    qq[..., 0] = 1.0
    kk[..., 0] = shift
    kk[0, 300] = qq[0, 5] * spike                    # query 5 (first query block of wave 0) in the 5th tile
    kk[0, 450] = qq[0, 45] * spike * 1.5             # query 45: the SECOND query block of wave 0
    kk[0, 639] = qq[0, 200] * spike * 2.0            # the very last key (the peeled last half step)
    kk[0, 35] = qq[0, 600] * spike                   # second half of the first tile
    qs, qeff = _prescaled_q(qq, d)
    kk = q(kk, dtype)
    ref = _ref_attn(qeff, kk, vv, heads)
    out = _run(dev, qs, kk, vv, heads, d, n, n)
    assert torch.isfinite(out).all()
    assert_close(out.float().cpu(), ref, dtype, what=f"flash v4 level move spike={spike} shift={shift}")


def test_v4_is_the_default_at_the_level0_shape(dev, monkeypatch):
    """(16, 8, 4096, 4096, 40) takes v4 by default and v3 with SASPA_ATTN_V4=0: the two differ in the last bit of some outputs only."""
    monkeypatch.delenv("SASPA_ATTN_MODE", raising=False)
    d, heads, bsz, n = 40, 8, 16, 4096
    g = torch.Generator(device="cpu").manual_seed(5)
    qs = (torch.randn(bsz, n, heads * d, generator=g) * 0.5).to(torch.bfloat16)
    kk = torch.randn(bsz, n, heads * d, generator=g).to(torch.bfloat16)
    vv = torch.randn(bsz, n, heads * d, generator=g).to(torch.bfloat16)
    monkeypatch.delenv("SASPA_ATTN_V4", raising=False)
    a = _run(dev, qs, kk, vv, heads, d, n, n)
    monkeypatch.setenv("SASPA_ATTN_V4", "0")
    b = _run(dev, qs, kk, vv, heads, d, n, n)
    assert torch.isfinite(a).all()
    diff = (a.float() - b.float()).abs().max().item()
    assert 0 < diff <= 2 ** -6 * b.float().abs().max().item() + 1e-3, diff      # (0 would mean the knob selected nothing)

"""UniPC sampler (`sampler="unipcmultistep"`, run_aug/run_aug.py:218-219).  CPU: the product's closed-form
per-step coefficient rows against the oracle's stateful restatement of UniPCMultistepScheduler on arbitrary model outputs,
timestep / sigma tables, the last step landing on the x0-prediction.  GPU: the fused CFG + UniPC kernel against the same
recurrence, and the whole pipeline (graph replay and Python launch loop) against the oracle pipeline with UniPC."""
import numpy as np
import pytest
import torch

import saspa_aug_amd  # noqa: F401
from oracle import pipeline as OP
from saspa_aug_amd.scheduler import DDIMScheduler, UniPCMultistepScheduler


def _recurrence(plan, x, eps_list):
    last, m0, m1 = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)
    for (t, r), e in zip(plan, eps_list):
        x0 = (x - r[1] * e) * r[0]
        if r[2]:
            x = r[3] * last + r[4] * m0 + r[5] * m1 + r[6] * x0
        m1, m0, last = m0, x0, x
        x = r[7] * x + r[8] * m0 + r[9] * m1
    return x, m0


@pytest.mark.parametrize("spacing", ["leading", "trailing"])
@pytest.mark.parametrize("steps", [1, 2, 3, 10, 30, 50])
def test_plan_equals_the_stateful_scheduler(steps, spacing):
    """"trailing" is what UniPCMultistepScheduler.from_config inherits from the sdxl-turbo scheduler config
    (run_aug/run_aug.py:223-226)."""
    sch = UniPCMultistepScheduler(timestep_spacing=spacing)
    plan = sch.plan(steps)
    o = OP.UniPC(spacing=spacing)
    ts = o.set_timesteps(steps)
    assert [t for t, _ in plan] == ts.tolist() and len(plan) == steps
    g = torch.Generator().manual_seed(steps)
    x = torch.randn(2, 4, 8, 8, generator=g)
    eps = [torch.randn(2, 4, 8, 8, generator=g) for _ in range(steps)]
    xo = x.clone()
    for e, t in zip(eps, ts):
        xo = o.step(e, t, xo)
    got, m0 = _recurrence(plan, x.double(), [e.double() for e in eps])
    assert (got.float() - xo).abs().max() < 1e-5 * xo.abs().max()
    assert torch.allclose(got, m0, rtol=1e-9, atol=1e-12)          # final sigma 0: the last step returns its x0-prediction


def test_tables_and_from_config():
    sch = UniPCMultistepScheduler.from_config(DDIMScheduler().config)
    ts = sch.set_timesteps(30)
    assert ts.tolist()[:3] == [961, 929, 897] and ts[-1] == 33 and len(ts) == 30          # leading: 1000 // (30 + 1) = 32, offset 1
    assert sch.sigmas.shape == (31,) and sch.sigmas[-1] == 0.0 and np.all(np.diff(sch.sigmas) < 0)
    with pytest.raises(NotImplementedError):
        UniPCMultistepScheduler(solver_order=3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_kernel(dev, dtype):
    from saspa_aug_amd import ops
    steps, b, hw = 6, 2, 64
    plan = UniPCMultistepScheduler().plan(steps)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(b, hw, 8, generator=g)
    x[..., 4:] = 0
    xd = torch.cat([x, x]).to(dev, dtype)
    state = torch.zeros((3, b, hw, 8), device=dev, dtype=dtype)
    table = torch.tensor([r for _, r in plan], dtype=torch.float32, device=dev)
    idx = torch.zeros(1, dtype=torch.int32, device=dev)
    ref = xd[:b].float().cpu()
    last, m0, m1 = torch.zeros_like(ref), torch.zeros_like(ref), torch.zeros_like(ref)
    for k, (t, r) in enumerate(plan):
        e2 = torch.randn(2 * b, hw, 8, generator=g).to(dtype)
        if k % 2 == 0:
            ops.cfg_unipc_step(e2.to(dev), xd, state, b, hw, 4, 7.5, row=r)
        else:
            idx.fill_(k)
            ops.cfg_unipc_step(e2.to(dev), xd, state, b, hw, 4, 7.5, table=table, index=idx)
        e = e2[:b].float() + 7.5 * (e2[b:].float() - e2[:b].float())
        q = (lambda v: v.to(dtype).float())
        x0 = (ref - r[1] * e) * r[0]
        xc = r[3] * last + r[4] * m0 + r[5] * m1 + r[6] * x0 if r[2] else ref
        m1, m0, last = m0, q(x0), q(xc)
        ref = q(r[7] * xc + r[8] * x0 + r[9] * m1)
        ref[..., 4:] = 0
        got = xd[:b].float().cpu()
        assert torch.equal(xd[:b], xd[b:])
        assert (got - ref).abs().max() < (1e-4 if dtype == torch.float32 else 6e-2) * max(1.0, ref.abs().max().item()), k


@pytest.mark.gpu
@pytest.mark.parametrize("graph", ["1", "0"])
def test_pipeline_fp32_parity_with_unipc(dev, graph, monkeypatch):
    from oracle.canny import generate_canny_array
    from saspa_aug_amd import config as CFG
    from saspa_aug_amd import run_aug as R
    from saspa_aug_amd import weights as W
    from saspa_aug_amd.synthetic import synthetic_image
    from tests.util import from_nhwc
    monkeypatch.setenv("SASPA_GRAPH", graph)
    cfgs = {k: v for k, v in CFG.tiny().items() if k != "safety"}
    fam = W.synth_family(cfgs, seed=3)
    pipe = R.init_pipeline("sd_v1.5", "canny", 0, sampler="unipcmultistep", cfgs=cfgs, state_dicts=fam)
    assert isinstance(pipe.scheduler, UniPCMultistepScheduler)
    pipe = pipe.to(dev, torch.float32)
    ids = np.random.RandomState(1).randint(0, cfgs["text"]["vocab"] - 2, (2, 77))
    neg = np.random.RandomState(2).randint(0, cfgs["text"]["vocab"] - 2, (1, 77))
    ctrls = np.stack([generate_canny_array(synthetic_image(64, 64, 10 + i), 120, 200) for i in range(2)])
    lat = torch.randn((2, 4, 8, 8), generator=torch.manual_seed(1))
    out, x, img = pipe.generate_batch(ids, neg, ctrls, lat, 8, return_latents=True)
    for i in range(2):
        ref_u8, ref_x, ref_img = OP.sd_controlnet_pipeline(fam, cfgs, torch.from_numpy(ids[i:i + 1]), torch.from_numpy(neg), ctrls[i],
                                                           lat[i:i + 1], 8, return_latents=True, sampler="unipc")
        d01 = ((from_nhwc(img[i:i + 1], 3) / 2 + 0.5).clamp(0, 1) - (ref_img / 2 + 0.5).clamp(0, 1)).abs().max().item()
        assert d01 < 1e-3 and np.abs(out[i:i + 1].cpu().numpy().astype(int) - ref_u8.astype(int)).max() <= 1, (i, d01)

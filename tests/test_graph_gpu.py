"""hipGraph replay of the DDIM sampling step (pipeline._StepGraph): the captured step must reproduce the eager launch
sequence bit for bit (same kernels, same order), across consecutive generations with different inputs (static buffers are
refreshed), different step counts / shapes (new cache keys) and for the CFG-free SDXL-Turbo form."""
import numpy as np
import pytest
import torch

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import config as CFG
from saspa_aug_amd import ops
from saspa_aug_amd import weights as W
from saspa_aug_amd.pipeline import (BlipDiffusionControlNetPipeline, StableDiffusionControlNetPipeline,
                                    StableDiffusionXLControlNetPipeline, graphs_enabled)
from saspa_aug_amd.synthetic import synthetic_image

pytestmark = pytest.mark.gpu


def _inputs(cfgs, n, hh, ww, seed):
    rs = np.random.RandomState(seed)
    ids = rs.randint(0, cfgs["text"]["vocab"] - 2, (n, 77))
    neg = rs.randint(0, cfgs["text"]["vocab"] - 2, (1, 77))
    ctrl = np.stack([(synthetic_image(hh, ww, seed + i) > 128).astype(np.uint8) * 255 for i in range(n)])
    lat = torch.randn((n, 4, hh // 8, ww // 8), generator=torch.manual_seed(seed))
    return ids, neg, ctrl, lat


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graph_replay_equals_eager_sd15(dev, dtype, monkeypatch):
    cfgs = CFG.tiny()
    fam = W.synth_family(cfgs, seed=3)
    pipe = StableDiffusionControlNetPipeline(fam, cfgs).to(dev, dtype)
    cases = [(2, 64, 64, 4, 11), (2, 64, 64, 4, 12), (1, 64, 128, 3, 13), (2, 64, 64, 6, 14), (2, 64, 64, 4, 15)]
    monkeypatch.setenv("SASPA_GRAPH", "0")
    assert not graphs_enabled()
    eager = [pipe.generate_batch(*_inputs(cfgs, n, hh, ww, seed), steps, return_latents=True)[1].clone() for (n, hh, ww, steps, seed) in cases]
    monkeypatch.setenv("SASPA_GRAPH", "1")
    assert graphs_enabled()
    for (n, hh, ww, steps, seed), ref in zip(cases, eager):
        got = pipe.generate_batch(*_inputs(cfgs, n, hh, ww, seed), steps, return_latents=True)[1]
        assert torch.equal(got, ref), f"graph replay differs from the eager loop for case {(n, hh, ww, steps, seed)}"
    assert 1 <= len(pipe._graphs) <= 3 and all(g.graph is not None for g in pipe._graphs.values())
    # a launch recorder (bench.py's roofline pass) turns the replay off
    ops.set_recorder(lambda kind, flops, call, meta=None: call())
    try:
        assert not graphs_enabled()
        got = pipe.generate_batch(*_inputs(cfgs, 2, 64, 64, 11), 4, return_latents=True)[1]
        # under a recorder every launch is timed alone and takes the full-chip split-K (no ops.twin_branch halving): the
        # same sums in another K order -- equal up to rounding, not bit for bit
        tol = 2e-5 if dtype == torch.float32 else 6e-2
        assert (got.float() - eager[0].float()).abs().max().item() <= tol * eager[0].float().abs().max().item()
    finally:
        ops.set_recorder(None)


def test_replay_throttle_and_clock_probe(dev, monkeypatch):
    """Round 6 host side: (a) the replay loop with at most 3 step-graph launches outstanding behind sleeping waits
    (SASPA_REPLAY_DEPTH, the default) produces the same latents as the unthrottled loop; (b) `saspa_clock_probe` measures a
    plausible shader clock (one sleeping wave: s_memtime against the 100 MHz s_memrealtime); (c) `ops.sleep_wait` returns once
    the event has completed."""
    cfgs = CFG.tiny()
    fam = W.synth_family(cfgs, seed=3)
    pipe = StableDiffusionControlNetPipeline(fam, cfgs).to(dev, torch.bfloat16)
    args = _inputs(cfgs, 2, 64, 64, 31)
    outs = {}
    for depth in ("3", "0", "1"):
        monkeypatch.setenv("SASPA_REPLAY_DEPTH", depth)
        outs[depth] = pipe.generate_batch(*args, 6, return_latents=True)[1].clone()
    assert torch.equal(outs["3"], outs["0"]) and torch.equal(outs["3"], outs["1"])
    buf = torch.zeros(2, dtype=torch.int64, device=dev)
    ops.clock_probe(buf, 250)
    ev = torch.cuda.Event()
    ev.record()
    ops.sleep_wait(ev)
    assert ev.query()
    clocks, ticks = (int(v) for v in buf.cpu())
    mhz = 100.0 * clocks / ticks
    assert ticks > 0 and 300.0 < mhz < 3500.0, (clocks, ticks, mhz)
    with pytest.raises(RuntimeError):
        ops.clock_probe(buf, 0)                                  # SASPA_ERANGE: a probe always terminates (1 .. 100000 windows)


def test_twin_recorder_charges_the_paired_region_its_wall_time(dev):
    """bench.Recorder(twin=True): the eager evaluation runs the two encoders on two streams (shared-chip dispatch) between
    begin_twin / end_twin, every launch is recorded (norms and element-wise passes too, through ops._RecordingLib), and the launches
    of a paired region are charged exactly that region's wall time in total."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import Recorder
    cfgs = CFG.tiny()
    fam = W.synth_family(cfgs, seed=3)
    pipe = StableDiffusionControlNetPipeline(fam, cfgs).to(dev, torch.bfloat16)
    args = _inputs(cfgs, 2, 64, 64, 41)
    ref = pipe.generate_batch(*args, 2, return_latents=True)[1].clone()
    rec = Recorder(twin=True)
    rec.calibrate(dev)
    ops.set_recorder(rec)
    try:
        got = pipe.generate_batch(*args, 2, return_latents=True)[1]
    finally:
        ops.set_recorder(None)
    # same dispatch as the captured step (twin hint on in the paired region): the same sums in the same order
    assert torch.equal(got, ref)
    ms = rec.charged_ms()
    assert len(ms) == len(rec.items) and all(m >= 0 for m in ms)
    kinds = {k for k, *_ in rec.items}
    assert {"gemm", "flash_attn", "groupnorm_apply", "layernorm"} <= kinds, kinds
    assert len(rec.twin_regions) == 2                               # one paired region per evaluation
    for g, reg in enumerate(rec.twin_regions):
        charged = sum(m for m, grp in zip(ms, rec.group) if grp == g)
        assert abs(charged - reg["wall_ms"]) <= 1e-3 * max(reg["wall_ms"], 1e-3) + 2e-3
    fams = {Recorder.kernel_family(k, m) for k, _, _, _, m in rec.items if k == "gemm"}
    assert fams <= {0, 1, 2, 3, 4, 8} and fams & {1, 2, 3, 4}


def test_graph_replay_equals_eager_sdxl(dev, monkeypatch):
    cfgs = CFG.tiny_xl()
    fam = W.synth_family(cfgs, seed=3)
    pipe = StableDiffusionXLControlNetPipeline(fam, cfgs).to(dev, torch.bfloat16)
    v = cfgs["text"]["vocab"]

    def inputs(n, seed):
        rs = np.random.RandomState(seed)
        ids = np.full((n, 77), v - 1, np.int64)
        ids[:, 0] = v - 2
        ids[:, 1:20] = rs.randint(0, v - 2, (n, 19))
        ctrl = np.stack([(synthetic_image(64, 64, seed + i) > 128).astype(np.uint8) * 255 for i in range(n)])
        return ids, None, ctrl, torch.randn((n, 4, 8, 8), generator=torch.manual_seed(seed))
    monkeypatch.setenv("SASPA_GRAPH", "0")
    eager = [pipe.generate_batch(*inputs(2, s), 2, return_latents=True)[1].clone() for s in (21, 22)]
    monkeypatch.setenv("SASPA_GRAPH", "1")
    for s, ref in zip((21, 22), eager):
        assert torch.equal(pipe.generate_batch(*inputs(2, s), 2, return_latents=True)[1], ref)


def test_step_state_kernels(dev):
    tab = torch.arange(5 * 12, device=dev, dtype=torch.float32).view(5, 12)
    idx = torch.tensor([3], device=dev, dtype=torch.int32)
    cur = torch.zeros(12, device=dev)
    ops.gather_row(tab, idx, cur)
    assert torch.equal(cur, tab[3])
    ops.index_add(idx, 1)
    assert idx.item() == 4
    # ddim_step_dev == cfg_ddim_step with the coefficients of the indexed row
    g = torch.Generator().manual_seed(0)
    eps = torch.randn(4, 6, 8, generator=g).to(dev)
    x = torch.randn(4, 6, 8, generator=g).to(dev)
    x[..., 4:] = 0
    coefs = torch.tensor([[0.9, 0.4, 0.95, 0.3], [0.8, 0.6, 0.85, 0.5]], device=dev)
    idx = torch.tensor([1], device=dev, dtype=torch.int32)
    a, b = x.clone(), x.clone()
    ops.cfg_ddim_step(eps, a, 2, 6, 4, 7.5, 0.8, 0.6, 0.85, 0.5)
    ops.ddim_step_dev(eps, b, 2, 6, 4, 7.5, coefs, idx, cfg=True)
    assert torch.equal(a, b)
    a, b = x.clone(), x.clone()
    ops.ddim_step(eps, a, 4, 6, 4, 0.8, 0.6, 0.85, 0.5)
    ops.ddim_step_dev(eps, b, 4, 6, 4, 0.0, coefs, idx, cfg=False)
    assert torch.equal(a, b)


def test_graph_replay_equals_eager_blip_plms(dev, monkeypatch):
    """PLMS (N + 1 evaluations, history ring, saved sample) through the captured step == the Python launch loop."""
    cfgs = CFG.tiny()
    fam = W.synth_family(cfgs, seed=3)
    pipe = BlipDiffusionControlNetPipeline(dict(fam), cfgs).to(dev, torch.bfloat16)
    nq, width = cfgs["qformer"]["num_query"], cfgs["text"]["width"]

    def inputs(n, seed):
        rs = np.random.RandomState(seed)
        ids = rs.randint(0, cfgs["text"]["vocab"] - 2, (n, pipe.prompt_token_count()))
        neg = rs.randint(0, cfgs["text"]["vocab"] - 2, (1, 77))
        ctrl = np.stack([(synthetic_image(64, 64, seed + i) > 128).astype(np.uint8) * 255 for i in range(n)])
        lat = torch.randn((n, 4, 8, 8), generator=torch.manual_seed(seed))
        q = torch.randn((n, nq, width), generator=torch.manual_seed(seed + 1)).to(dev)
        return (ids, neg, ctrl, lat, 6, 7.5, 1.0), dict(query_embeds=q, return_latents=True)
    monkeypatch.setenv("SASPA_GRAPH", "0")
    eager = []
    for s_ in (31, 32):
        a, kw = inputs(2, s_)
        eager.append(pipe.generate_batch(*a, **kw)[1].clone())
    monkeypatch.setenv("SASPA_GRAPH", "1")
    for s_, ref in zip((31, 32), eager):
        a, kw = inputs(2, s_)
        assert torch.equal(pipe.generate_batch(*a, **kw)[1], ref)


def test_plms_step_dev_kernel(dev):
    g = torch.Generator().manual_seed(0)
    eps = torch.randn(4, 6, 8, generator=g).to(dev)
    x = torch.randn(4, 6, 8, generator=g).to(dev)
    x[..., 4:] = 0
    x[2:] = x[:2]
    hist = torch.randn(4, 2, 6, 8, generator=g).to(dev)
    saved = torch.randn(2, 6, 8, generator=g).to(dev)
    # row 0: store into slot 2, two history weights, read the saved sample; row 1: save the sample, no store
    table = torch.tensor([[2, 1.5, 0.0, -0.5, 0.0, 0.25, 0.9, -0.2, 0, 1], [-1, 1.0, 0, 0, 0, 0, 0.8, -0.1, 1, 0]], device=dev,
                         dtype=torch.float32)
    a, ha = x.clone(), hist.clone()
    ops.cfg_plms_step(eps, a, ha, saved, 2, 6, 4, 7.5, 2, 1.5, [0.0, -0.5, 0.0, 0.25], 0.9, -0.2)
    b, hb, sb = x.clone(), hist.clone(), saved.clone()
    ops.cfg_plms_step_dev(eps, b, hb, sb, 2, 6, 4, 7.5, table, torch.tensor([0], device=dev, dtype=torch.int32))
    assert torch.equal(a, b) and torch.equal(ha, hb) and torch.equal(sb, saved)
    a, ha = x.clone(), hist.clone()
    ops.cfg_plms_step(eps, a, ha, None, 2, 6, 4, 7.5, -1, 1.0, [0.0] * 4, 0.8, -0.1)
    b, hb, sb = x.clone(), hist.clone(), saved.clone()
    ops.cfg_plms_step_dev(eps, b, hb, sb, 2, 6, 4, 7.5, table, torch.tensor([1], device=dev, dtype=torch.int32))
    assert torch.equal(a, b) and torch.equal(ha, hb) and torch.equal(sb, x[:2])

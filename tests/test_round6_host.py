"""Round-6 host-side pieces that need no GPU: the pre-split SASPA_F32X3 weight layout, the fp8 tensor scale, the CPU-pool sizing of
a generation rank, the launch recorder's bookkeeping (twin regions, refused probes, the recording proxy)."""
import math
import os

import pytest
import torch

import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
from saspa_aug_amd import run_aug as R
from saspa_aug_amd import weights as W


def test_presplit_x3_layout_and_values():
    """weights.presplit_x3 (SaspaGemmParams.w_split, ABI 20): per K-tile of 32 values [32 bf16 hi | 32 bf16 lo], hi = bf16(w) (RNE),
    lo = bf16(w - hi); 16-byte chunk c of either half carries tile positions 4c..4c+3 and 16+4c..16+4c+3 -- the fp32 chunks c and
    4 + c a lane group reads of the activation tile."""
    w = torch.randn(7, 96, generator=torch.Generator().manual_seed(0)) * torch.logspace(-3, 2, 96)
    o = W.presplit_x3(w)
    assert o.shape == w.shape and o.dtype == torch.float32
    b = o.view(torch.bfloat16).reshape(7, 3, 64)
    for c in range(4):
        pos = list(range(4 * c, 4 * c + 4)) + list(range(16 + 4 * c, 20 + 4 * c))
        src = w.reshape(7, 3, 32)[:, :, pos]
        hi = b[:, :, 8 * c:8 * c + 8]
        lo = b[:, :, 32 + 8 * c:32 + 8 * c + 8]
        assert torch.equal(hi, src.to(torch.bfloat16))
        assert torch.equal(lo, (src - hi.float()).to(torch.bfloat16))
        # the pair carries ~16 mantissa bits of the weight
        assert ((hi.float() + lo.float() - src).abs() <= 2.0 ** -16 * src.abs() + 1e-30).all()
    with pytest.raises(ValueError):
        W.presplit_x3(torch.randn(4, 40))
    with pytest.raises(ValueError):
        W.presplit_x3(torch.randn(4, 64).bfloat16())


def test_fp8_pow2_scale():
    """ops.fp8_pow2_scale: the power of two >= margin * amax / 448; 1 for an all-zero tensor; no host sync (tensor in, tensor out)."""
    for amax in (1e-6, 0.37, 1.0, 448.0 / 16, 100.0, 3.2e4):
        s = ops.fp8_pow2_scale(torch.tensor([amax])).item()
        assert math.log2(s) == round(math.log2(s))
        assert 16 * amax / 448 <= s < 2 * 16 * amax / 448 * (1 + 1e-6)
    assert ops.fp8_pow2_scale(torch.zeros(1)).item() == 1.0
    assert ops.fp8_pow2_scale(torch.tensor([1.0]), margin=1.0).item() == 2.0 ** -8     # 1 / 448 -> next power of two 1 / 256


def test_host_threads(monkeypatch):
    monkeypatch.delenv("SASPA_HOST_THREADS", raising=False)
    n = R.host_threads(1)
    assert 1 <= n <= 4
    assert R.host_threads(10 ** 6) == 1                      # more ranks than cores: never below one thread
    monkeypatch.setenv("SASPA_HOST_THREADS", "7")
    assert R.host_threads(1) == 7 and R.host_threads(64) == 7


def test_main_restores_the_cpu_pool(monkeypatch):
    """run_aug.main holds torch's CPU pool at host_threads() for the run and puts the previous size back, also when the run raises."""
    before = torch.get_num_threads()
    seen = []

    def fake_main(*a, **k):
        seen.append(torch.get_num_threads())
        raise RuntimeError("boom")
    monkeypatch.setattr(R, "_main", fake_main)
    monkeypatch.setenv("SASPA_HOST_THREADS", "2")
    with pytest.raises(RuntimeError):
        R.main(R.Settings(DATASET="synthetic"))
    assert seen == [2] and torch.get_num_threads() == before


class _FakeLib:
    def __init__(self):
        self.calls = []

    def saspa_layernorm(self, *a):
        self.calls.append("layernorm")
        return 0

    def saspa_gemm_which(self, *a):
        self.calls.append("which")
        return 2 | (3 << 8)

    def saspa_groupnorm_onepass_eligible(self, *a):
        return 1


def test_recording_proxy_and_probe_launch(monkeypatch):
    """ops._RecordingLib times every kernel-launching entry point that `_launch` does not already wrap, never the host-side
    queries; a refused probe launch (SASPA_ERANGE) is not recorded; `_launch` inside a recorded call is not recorded twice."""
    rec_log = []

    class Rec:
        def __call__(self, kind, flops, call, meta=None):
            rec_log.append((kind, flops, meta))
            return call()

        def conditional(self, kind, flops, call, meta, keep):
            r = call()
            if keep(r):
                rec_log.append((kind, flops, meta))
            return r
    fake = _FakeLib()
    proxy = ops._RecordingLib(fake)
    monkeypatch.setattr(ops, "_RECORDER", Rec())
    assert proxy.saspa_layernorm(1, 2) == 0 and rec_log == [("layernorm", 0.0, None)]
    assert proxy.saspa_gemm_which(1) == (2 | (3 << 8)) and len(rec_log) == 1                 # host-side query: not recorded
    assert proxy.saspa_groupnorm_onepass_eligible(1) == 1 and len(rec_log) == 1
    # a launch that is already inside `_launch` goes through the proxy unrecorded (one entry per launch)
    ops._launch("gemm", 10.0, lambda: proxy.saspa_layernorm(3), ("m",))
    assert [k for k, *_ in rec_log] == ["layernorm", "gemm"]
    # refused probe: nothing recorded; accepted probe: recorded once
    assert ops._probe_launch("gemm", 5.0, lambda: -3, ("x",)) == -3 and len(rec_log) == 2
    assert ops._probe_launch("gemm", 5.0, lambda: 0, ("y",)) == 0 and rec_log[-1] == ("gemm", 5.0, ("y",))
    assert ops._IN_LAUNCH[0] is False
    monkeypatch.setattr(ops, "_RECORDER", None)
    assert ops._probe_launch("gemm", 5.0, lambda: 7) == 7 and len(rec_log) == 3


def test_recorder_twin_charging_arithmetic():
    """bench.Recorder: the launches of a twin region are charged duration x (region wall / sum of durations) -- the region's charges
    add up to its wall time; launches outside keep their own duration; the event-pair floor is subtracted first.  (Pure arithmetic,
    with stand-in events.)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t
    rec = bench.Recorder(twin=True)
    rec.floor_ms = 0.01
    rec.items = [("gemm", 1.0, Ev(0.0), Ev(1.01), None),                    # outside: 1.0 after the floor
                 ("gemm", 1.0, Ev(2.0), Ev(4.01), None), ("gemm", 1.0, Ev(2.5), Ev(3.51), None),   # a region: wall 2.01, durations 2.0 + 1.0
                 ("layernorm", 0.0, Ev(5.0), Ev(5.21), None)]
    rec.group = [None, 0, 0, None]
    rec._n_groups = 1
    import torch as _t
    real_sync = _t.cuda.synchronize
    _t.cuda.synchronize = lambda *a, **k: None
    try:
        ms = rec.charged_ms()
    finally:
        _t.cuda.synchronize = real_sync
    assert abs(ms[0] - 1.0) < 1e-9 and abs(ms[3] - 0.2) < 1e-9
    assert abs((ms[1] + ms[2]) - 2.01) < 1e-9 and abs(ms[1] / ms[2] - 2.0) < 1e-9
    assert rec.twin_regions == [dict(launches=2, wall_ms=2.01, sum_of_durations_ms=3.0)]


def test_ff_block_takes_whole_rounds(monkeypatch):
    """models.ff_block_takes: the one-launch feed-forward is taken when its 128-row workgroups fill their last round of 256 CUs
    (tools/ff_bench.py m=...: wins at 24 576 / 32 768 / 65 536 / 90 112 / 98 304 rows, loses at 16 384 and at 49 152 = 1.5 rounds)."""
    from saspa_aug_amd import models
    monkeypatch.delenv("SASPA_FF_BLOCK_MIN_ROWS", raising=False)
    takes = {rows: models.ff_block_takes(rows) for rows in (8192, 16384, 24576, 32768, 49152, 65536, 90112, 98304, 106496, 114688)}
    assert takes == {8192: False, 16384: False, 24576: True, 32768: True, 49152: False, 65536: True, 90112: True, 98304: True,
                     106496: True, 114688: True}, takes
    monkeypatch.setenv("SASPA_FF_BLOCK_MIN_ROWS", "0")
    assert models.ff_block_takes(128) and models.ff_block_takes(49152)
    monkeypatch.setenv("SASPA_FF_BLOCK_MIN_ROWS", "70000")
    assert not models.ff_block_takes(65536) and models.ff_block_takes(90112)


def test_noise_replay_skips_other_ranks_items_bit_exactly(monkeypatch):
    """run_aug.noise_for_items advances the CPU generator past the items of other ranks with byte draws instead of drawing their noise
    (rank 7 spent 10.5 s replaying 11 669 items).  The skip is self-checked (`_noise_skip_is_exact`) and must give,
    for every rank, exactly the tensors of the sequential stream -- one and two draws per item, fp16 and fp32, mixed bucket sizes."""
    import torch
    from saspa_aug_amd import run_aug as R
    sizes = [(512, 512), (512, 704), (512, 768), (576, 512), (512, 896)]
    items = [R.WorkItem(k, k, "", "", 0, "", "", sizes[k % 5][0], sizes[(k * 7) % 5][1]) for k in range(41)]
    items[4].skip = items[17].skip = True
    for dtype in (torch.float16, torch.float32):
        assert R._noise_skip_is_exact(dtype)                      # holds on this torch: the fast path is the one under test
        for draws in (1, 2):
            g = torch.manual_seed(7)
            ref = {it.order: torch.cat([torch.randn((1, 4, it.height // 8, it.width // 8), generator=g, dtype=dtype) for _ in range(draws)])
                   for it in items if not it.skip}
            for sh in R.shard_items(items, 8):
                monkeypatch.setenv("SASPA_NOISE_SKIP", "1")
                fast = R.noise_for_items(items, sh, 7, dtype, draws)
                monkeypatch.setenv("SASPA_NOISE_SKIP", "0")
                slow = R.noise_for_items(items, sh, 7, dtype, draws)
                assert set(fast) == set(slow) == {it.order for it in sh}
                for k in fast:
                    assert torch.equal(fast[k], ref[k]) and torch.equal(slow[k], ref[k])
    # a tensor whose element count is not a multiple of 16 is drawn, not skipped (torch's scalar tail path consumes differently)
    odd = [R.WorkItem(k, k, "", "", 0, "", "", 24, 40) for k in range(6)]          # 4 * 3 * 5 = 60 elements
    g = torch.manual_seed(3)
    ref = [torch.randn((1, 4, 3, 5), generator=g, dtype=torch.float32) for _ in range(6)]
    monkeypatch.setenv("SASPA_NOISE_SKIP", "1")
    got = R.noise_for_items(odd, odd[4:], 3, torch.float32)
    assert torch.equal(got[4], ref[4]) and torch.equal(got[5], ref[5])

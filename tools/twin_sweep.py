"""Kernel choice for the TWIN region of a sampling step (UNet encoder || ControlNet encoder: two graph branches that walk the
same layer shapes side by side, pipeline._StepGraph): every shape is launched on TWO streams at once -- each stream its own
activations, weights and outputs -- and timed as a pair, for AUTO (the single-branch dispatch), AUTO with the twin split-K
halving (ops.twin_branch) and the pinned kernels with explicit K slices.  What is printed is the time of the PAIR per
iteration (us) and the combined TFLOP/s.  Decides the twin-aware dispatch rules from data.
usage: python tools/twin_sweep.py [conv|linear|all]"""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops, weights as W
dev = torch.device('cuda:0'); BF = torch.bfloat16
what = sys.argv[1] if len(sys.argv) > 1 else "all"
CONVS = [(16, 32, 32, 640, 640), (16, 16, 16, 1280, 1280), (16, 8, 8, 1280, 1280), (16, 32, 32, 320, 640), (16, 16, 16, 640, 1280),
         (16, 64, 64, 320, 320)]
LINS = [  # (M, N, K, residual, geglu)
    (16384, 640, 640, True, False), (16384, 1920, 640, False, False), (16384, 5120, 640, False, True), (16384, 640, 2560, True, False),
    (4096, 1280, 1280, True, False), (4096, 3840, 1280, False, False), (4096, 10240, 1280, False, True), (4096, 1280, 5120, True, False),
    (1024, 1280, 1280, True, False), (1024, 1280, 5120, True, False), (1024, 10240, 1280, False, True), (65536, 320, 1280, True, False),
    # the 512x704 bucket's level 1 (round 6)
    (22528, 640, 640, True, False), (22528, 1920, 640, False, False), (22528, 5120, 640, False, True), (22528, 640, 2560, True, False)]
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def time_pair(f1, f2, n=20):
    """f1 on stream s1 and f2 on stream s2, n iterations each, issued interleaved; wall time of both per iteration."""
    for _ in range(40):          # warm: clocks drop while the host builds the next shape's operands, lazy code-object loads
        with torch.cuda.stream(s1): f1()
        with torch.cuda.stream(s2): f2()
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    s1.wait_event(e0); s2.wait_event(e0)
    for _ in range(n):
        with torch.cuda.stream(s1): f1()
        with torch.cuda.stream(s2): f2()
    e1.record(s1); e2.record(s2)
    torch.cuda.synchronize()
    return max(e0.elapsed_time(e1), e0.elapsed_time(e2)) / n * 1e3


def time_single(f, n=20):
    for _ in range(40): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(label, flops, make):
    """make(variant, ks, twin) -> (f1, f2): two independent launch closures."""
    row = []
    cases = [("single x2", 0, None, False, True), ("auto", 0, None, False, False), ("auto+twin", 0, None, True, False),
             ("tiled k1", 1, 1, False, False), ("tiled k2", 1, 2, False, False), ("wide k1", 2, 1, False, False),
             ("wide k2", 2, 2, False, False), ("ws", 3, 1, False, False), ("auto (again)", 0, None, False, False)]
    for (name, variant, ks, twin, single) in cases:
        try:
            f1, f2 = make(variant, ks, twin)
            if single:       # the pair run back to back on ONE stream (what the single-branch order costs)
                us = time_single(lambda: (f1(), f2()))
            else:
                us = time_pair(f1, f2)
            row.append(f"{name} {us:6.1f}us {2 * flops / us / 1e6:5.0f}TF")
        except RuntimeError:
            row.append(f"{name} n/a")
    print(f"{label}: " + " | ".join(row), flush=True)


if what in ("conv", "all"):
    for (b, h, w_, cin, cout) in CONVS:
        M, K = b * h * w_, 9 * cin
        xs = [[torch.randn(b, h, w_, cin, device=dev).to(BF) for _ in range(3)] for _ in range(2)]
        wts = []
        for _ in range(2):
            wt = W.to_chunk_major((torch.randn(cout, K) / math.sqrt(K)), 9, BF).to(dev, BF); wt.saspa_korder = 1
            wts.append(wt)
        bias = torch.randn(cout, device=dev)
        cnt = [0, 0]

        def make(variant, ks, twin):
            def mk(side):
                def f():
                    j = cnt[side] % 3; cnt[side] += 1
                    with ops.twin_branch(twin):
                        ops.conv(xs[side][j], wts[side], bias, kh=3, kw=3, pad=1, variant=variant, ksplit=ks, gn_unit=None)
                return f
            return mk(0), mk(1)
        run(f"conv M={M} N={cout} K={K}", 2.0 * M * cout * K, make)

if what in ("linear", "all"):
    for (M, N, K, has_res, geglu) in LINS:
        xs = [[torch.randn(M, K, device=dev).to(BF) for _ in range(3)] for _ in range(2)]
        ress = [torch.randn(M, N, device=dev).to(BF) if has_res else None for _ in range(2)]
        wts, biases = [], []
        for _ in range(2):
            wf, bf = torch.randn(N, K) / math.sqrt(K), torch.randn(N)
            if geglu:
                wf, bf = W.pack_geglu(wf, bf)
            wts.append(wf.to(dev, BF)); biases.append(bf.to(dev))
        cnt = [0, 0]

        def make(variant, ks, twin):
            def mk(side):
                def f():
                    j = cnt[side] % 3; cnt[side] += 1
                    with ops.twin_branch(twin):
                        ops.linear(xs[side][j], wts[side], biases[side], residual=ress[side], act=ops.ACT_GEGLU if geglu else ops.ACT_NONE,
                                   variant=variant, ksplit=(None if geglu else ks))
                return f
            return mk(0), mk(1)
        run(f"linear M={M} N={N} K={K}{' +res' if has_res else ''}{' geglu' if geglu else ''}", 2.0 * M * N * K, make)

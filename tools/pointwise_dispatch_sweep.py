"""Pointwise layer shapes of a UNet + ControlNet evaluation as the pipeline launches them (bias, with / without residual): what AUTO
picks against each pinned kernel family (tiled = 4-wave 128x160, wide = 8-wave 256x320, ws = wave-specialised), library-chosen
K-split everywhere.  Re-run after a kernel's fixed cost changes: the dispatch thresholds were fitted on the costs of their time."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401,E402
from saspa_aug_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
BF = torch.bfloat16
# (M, N, K, residual)
shapes = [(65536, 960, 320, False), (65536, 640, 320, False), (65536, 320, 320, True), (65536, 320, 1280, True),
          (16384, 1920, 640, False), (16384, 640, 640, True), (16384, 640, 640, False), (16384, 640, 2560, True), (16384, 640, 1280, True), (16384, 640, 960, True),
          (4096, 3840, 1280, False), (4096, 1280, 1280, True), (4096, 1280, 1280, False), (4096, 1280, 5120, True), (4096, 1280, 2560, True), (4096, 1280, 1920, True),
          (1024, 1280, 1280, True), (1024, 1280, 2560, True),
          # the 512x704 bucket's levels (round 6)
          (22528, 1920, 640, False), (22528, 640, 640, True), (22528, 640, 2560, True), (5632, 3840, 1280, False), (5632, 1280, 1280, True)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for (M, N, K, has_res) in shapes:
    xs = [torch.randn(M, K, device=dev).to(BF) for _ in range(3)]
    wt = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).to(BF) if has_res else None
    outs = [torch.empty(M, N, device=dev, dtype=BF) for _ in range(3)]
    i = [0]
    row = []
    for (name, variant) in (("auto", 0), ("tiled", 1), ("wide", 2), ("ws", 3)):
        def f():
            j = i[0] % 3
            i[0] += 1
            ops.linear(xs[j], wt, bias, residual=res, out=outs[j], variant=variant)
        try:
            us = timeit(f)
            row.append(f"{name} {us:6.1f}")
        except RuntimeError:
            row.append(f"{name}    n/a")
    print(f"M={M:6d} N={N:5d} K={K:5d} res={int(has_res)}: " + " | ".join(row) + "  us", flush=True)

# fused-GEGLU projections (ff.net.0.proj, weights regrouped per output tile; random weights time the same): AUTO against the pinned
# kernels.  Round 6: re-run after the 8-wave kernel's long-interval loop (+2 ... +8 % per launch) -- does its K >= 1024 gate still hold?
print("GEGLU projections:")
for (M, N, K) in [(65536, 2560, 320), (16384, 5120, 640), (4096, 10240, 1280), (22528, 5120, 640), (5632, 10240, 1280)]:
    xs = [torch.randn(M, K, device=dev).to(BF) for _ in range(3)]
    wt = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
    bias = torch.randn(N, device=dev)
    outs = [torch.empty(M, N // 2, device=dev, dtype=BF) for _ in range(3)]
    i = [0]
    row = []
    for (name, variant) in (("auto", 0), ("tiled", 1), ("wide", 2), ("ws", 3), ("as", 4)):
        def f():
            j = i[0] % 3
            i[0] += 1
            ops.linear(xs[j], wt, bias, act=ops.ACT_GEGLU, out=outs[j], variant=variant)
        try:
            us = timeit(f)
            row.append(f"{name} {us:6.1f} ({2.0 * M * N * K / us / 1e6:5.0f} TF/s)")
        except (RuntimeError, ValueError):
            row.append(f"{name}    n/a")
    print(f"M={M:6d} N={N:5d} K={K:5d}: " + " | ".join(row), flush=True)

"""Fixed cost per launch of the 8-wave conv kernel: T(K) = a + b K on the 3x3 conv shapes of the three UNet levels.

For each level (M rows, N channels) the conv is timed at K = 9 C for C = N, 2 N, 3 N (what the decoder's concatenated inputs
give), each with the bare epilogue and with the residual + GroupNorm-statistics epilogue; a least-squares line through the three
K gives the per-launch fixed part `a` (launch + prologue + epilogue + tail) and the MFMA rate of the K loop from `b`.
Timed in batches of back-to-back launches between HIP events (the host enqueues faster than the kernels run)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401,E402
from saspa_aug_amd import ops, weights as W  # noqa: E402

dev = torch.device('cuda:0')
BF = torch.bfloat16


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


print(f"{'level':22s} {'epilogue':16s} " + " ".join(f"{'K=' + str(9 * c) + 'N/N':>10s}" for c in (1, 2, 3)) + f" {'a (us)':>8s} {'b (us/9N)':>10s} {'loop TF/s':>10s} {'a / T(K=9N)':>12s}")
for (b, h, n) in ((16, 64, 320), (16, 32, 640), (16, 16, 1280), (16, 8, 1280)):
    m = b * h * h
    for epi in ("bare", "res+stats", "bias+temb+stats"):        # conv2 (+ bias + shortcut) / conv1 (+ bias + time embedding)
        ts = []
        for cm in (1, 2, 3):
            c = n * cm
            x = torch.randn(b, h, h, c, device=dev).to(BF)
            wt = W.to_chunk_major(torch.randn(n, 9 * c) / math.sqrt(9 * c), 9, BF).to(dev, BF)
            wt.saspa_korder = 1
            res = torch.randn(b, h, h, n, device=dev).to(BF) if epi == "res+stats" else None
            out = torch.empty(b, h, h, n, device=dev, dtype=BF)
            bias = torch.randn(n, device=dev) if epi != "bare" else None
            rv = torch.randn(b, n, device=dev) if epi == "bias+temb+stats" else None
            f = lambda: ops.conv(x, wt, bias, kh=3, kw=3, pad=1, residual=res, rowvec=rv, out=out, gn_unit=(n // 32 if epi != "bare" else None))
            ts.append(timeit(f))
            del x, wt
        # least squares through (1, 2, 3)
        xs = [1.0, 2.0, 3.0]
        mx, my = 2.0, sum(ts) / 3
        bb = sum((xi - mx) * (ti - my) for xi, ti in zip(xs, ts)) / 2.0
        aa = my - bb * mx
        tf = 2.0 * m * n * 9 * n / (bb * 1e-6) / 1e12
        print(f"{m:6d} x {n:4d} (h={h:2d})   {epi:16s} " + " ".join(f"{t:10.1f}" for t in ts) + f" {aa:8.1f} {bb:10.1f} {tf:10.0f} {aa / ts[0]:12.2f}")

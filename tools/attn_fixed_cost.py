"""Fixed cost per launch of the level-0 flash attention (16 x 8 heads x 4096 queries, d = 40): T(nk) = a + b nk over
nk = 1024 ... 8192 keys (the production launch is nk = 4096), production flags (prescaled queries, V^T operand)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401,E402
from saspa_aug_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B, H, NQ, D = 16, 8, 4096, 40
C = H * D


def timeit(fn, n=20):
    for _ in range(3):
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


q = torch.randn(B, NQ, C, device=dev).bfloat16()
out = torch.empty(B, NQ, C, device=dev, dtype=torch.bfloat16)
xs, ts = [], []
for nk in (1024, 2048, 4096, 8192):
    k = torch.randn(B, nk, C, device=dev).bfloat16()
    vt = torch.randn(B, C, nk, device=dev).bfloat16()
    t = timeit(lambda: ops.flash_attn(q, k, vt, out, H, D, NQ, nk, D ** -0.5, prescaled=True))
    xs.append(nk / 1024.0)
    ts.append(t)
    print(f"nk = {nk:5d}: {t:8.1f} us   {4.0 * B * H * NQ * nk * D / t / 1e6:7.1f} TFLOP/s")
mx, my = sum(xs) / len(xs), sum(ts) / len(ts)
b = sum((x - mx) * (t - my) for x, t in zip(xs, ts)) / sum((x - mx) ** 2 for x in xs)
a = my - b * mx
print(f"T = {a:.1f} us + {b:.1f} us per 1024 keys  ->  loop rate {4.0 * B * H * NQ * 1024 * D / b / 1e6:.0f} TFLOP/s, fixed part {a / ts[2] * 100:.0f} % of the nk = 4096 launch")

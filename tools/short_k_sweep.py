"""Short-K linear shapes of the transformer blocks (K = 320 / 640 / 1280): 4-wave 128x160 tiles vs the 8-wave 256x320 wide
kernel (pinned through SaspaGemmParams.variant), plain epilogue (bias only) -- decides the dispatch gate and whether a
GEGLU epilogue in the wide kernel is worth building."""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
dev = torch.device('cuda:0'); BF = torch.bfloat16
shapes = [(65536, 2560, 320), (16384, 5120, 640), (4096, 10240, 1280), (65536, 320, 320), (65536, 640, 320), (16384, 640, 640),
          (16384, 1280, 640), (4096, 1280, 1280), (4096, 2560, 1280), (65536, 320, 1280), (16384, 640, 2560), (4096, 1280, 5120)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in shapes:
    xs = [torch.randn(M, K, device=dev).to(BF) for _ in range(4)]
    wt = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
    bias = torch.randn(N, device=dev)
    outs = [torch.empty(M, N, device=dev, dtype=BF) for _ in range(4)]
    i = [0]; row = []
    for (name, variant) in (("auto", 0), ("tiled", 1), ("wide", 2), ("ws", 3)):
        def f():
            j = i[0] % 4; i[0] += 1
            ops.linear(xs[j], wt, bias, out=outs[j], variant=variant, ksplit=1)
        try:
            us = timeit(f)
            row.append(f"{name} {us:6.1f}us {2.0 * M * N * K / us / 1e6:5.0f}TF")
        except RuntimeError as e:
            row.append(f"{name} n/a")
    print(f"M={M} N={N} K={K}: " + " | ".join(row), flush=True)

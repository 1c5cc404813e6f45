"""Linear-layer shapes with K >= 960: AUTO vs pinned kernels with explicit split-K (see conv_variant_sweep.py)."""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
dev = torch.device('cuda:0'); BF = torch.bfloat16
shapes = [(4096, 1280, 5120), (16384, 640, 2560), (4096, 1280, 1280), (16384, 1280, 640), (4096, 2560, 1280), (1024, 1280, 5120), (1024, 1280, 1280),
          (8192, 1280, 5120), (2056, 1024, 4096), (2056, 4096, 1024)]
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
    res = torch.randn(M, N, device=dev).to(BF)
    wt = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
    bias = torch.randn(N, device=dev)
    i = [0]; row = []
    for (name, variant, ks) in (("auto", 0, None), ("tiled k1", 1, 1), ("tiled k2", 1, 2), ("tiled k4", 1, 4), ("wide k1", 2, 1), ("wide k2", 2, 2),
                                ("wide k4", 2, 4), ("wide k8", 2, 8)):
        def f():
            j = i[0] % 4; i[0] += 1
            ops.linear(xs[j], wt, bias, residual=res, variant=variant, ksplit=ks)
        try:
            us = timeit(f)
            row.append(f"{name} {us:6.1f}us {2.0 * M * N * K / us / 1e6:5.0f}TF")
        except RuntimeError:
            row.append(f"{name} n/a")
    print(f"M={M} N={N} K={K}: " + " | ".join(row), flush=True)

"""3x3 conv shapes of the deep UNet levels: AUTO dispatch vs the pinned kernels with explicit split-K factors (HIP events,
distinct input buffers per call).  Decides the dispatch / split-K heuristics from data."""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops, weights as W
dev = torch.device('cuda:0'); BF = torch.bfloat16
shapes = [(16, 16, 16, 1280, 1280), (16, 8, 8, 1280, 1280), (16, 32, 32, 640, 640), (16, 16, 16, 2560, 1280), (16, 32, 32, 1280, 640),
          (16, 8, 8, 2560, 1280), (16, 32, 32, 960, 640), (16, 16, 16, 1920, 1280)]
if "704" in sys.argv:      # the 512x704 bucket of BASELINE configs[3] (latents 64x88): tile counts that are no multiple of the CU count
    shapes = [(16, 64, 88, 320, 320), (16, 64, 88, 640, 320), (16, 32, 44, 640, 640), (16, 32, 44, 1280, 640), (16, 32, 44, 1920, 640),
              (16, 16, 22, 1280, 1280), (16, 16, 22, 2560, 1280), (16, 16, 22, 1920, 1280), (16, 8, 11, 1280, 1280), (16, 8, 11, 2560, 1280)]
if "768" in sys.argv:
    shapes = [(16, 64, 96, 320, 320), (16, 32, 48, 640, 640), (16, 32, 48, 1280, 640), (16, 16, 24, 1280, 1280), (16, 16, 24, 2560, 1280),
              (16, 8, 12, 1280, 1280)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (b, h, w_, cin, cout) in shapes:
    M, K = b * h * w_, 9 * cin
    xs = [torch.randn(b, h, w_, cin, device=dev).to(BF) for _ in range(4)]
    wt = W.to_chunk_major((torch.randn(cout, K) / math.sqrt(K)), 9, BF).to(dev, BF); wt.saspa_korder = 1
    bias = torch.randn(cout, device=dev)
    i = [0]
    row = []
    cases = (("auto", 0, None), ("tiled k1", 1, 1), ("tiled k2", 1, 2), ("tiled k4", 1, 4), ("tiled k8", 1, 8),
             ("wide k1", 2, 1), ("wide k2", 2, 2), ("wide k4", 2, 4), ("wide k8", 2, 8))
    if "704" in sys.argv or "768" in sys.argv or "all" in sys.argv:
        cases = (("auto", 0, None), ("tiled k1", 1, 1), ("tiled k2", 1, 2), ("tiled k3", 1, 3), ("tiled k4", 1, 4), ("tiled k6", 1, 6)) + \
            tuple((f"wide k{k}", 2, k) for k in (1, 2, 3, 4, 5, 6, 7, 8))
    cases = cases + (("auto (again)", 0, None),)      # the first case of a row runs on clocks that dropped while the host built the weights
    for (name, variant, ks) in cases:
        def f():
            j = i[0] % 4; i[0] += 1
            ops.conv(xs[j], wt, bias, kh=3, kw=3, pad=1, variant=variant, ksplit=ks)
        try:
            us = timeit(f)
            row.append(f"{name} {us:6.1f}us {2.0 * M * cout * K / us / 1e6:6.0f}TF")
        except RuntimeError as e:
            row.append(f"{name} n/a")
    print(f"M={M} N={cout} K={K}: " + " | ".join(row), flush=True)

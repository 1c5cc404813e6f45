"""Fixed cost per launch of the pointwise (1x1 / linear) GEMMs of the 32x32 and 16x16 levels: T(K) = a + b K per kernel variant
at M = 16384, N = 640 and M = 4096, N = 1280 (the attention / feed-forward output projections and 1x1 convs: ~100 launches per
UNet + ControlNet evaluation at ~30 us), with and without the residual epilogue.  usage: python tools/pw_fixed_cost.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401,E402
from saspa_aug_amd import ops  # noqa: E402
dev = torch.device('cuda:0'); BF = torch.bfloat16


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


VARIANTS = [("auto", ops.GEMM_AUTO), ("tiled", ops.GEMM_TILED), ("ws", ops.GEMM_WS), ("wide", ops.GEMM_WIDE)]
for (m, n) in ((16384, 640), (4096, 1280), (65536, 320)):
    print(f"M = {m}, N = {n}: us per launch (res = with residual epilogue); fit T = a + b * (K / 64)")
    print(f"{'K':>6s} " + " ".join(f"{v + ' ':>9s}{v + '+res':>10s}" for v, _ in VARIANTS))
    rows = {}
    for k in (64, 128, 320, 640, 1280, 2560):
        xs = [torch.randn(m, k, device=dev).to(BF) for _ in range(3)]
        w = (torch.randn(n, k, device=dev) / k ** 0.5).to(BF)
        bias = torch.randn(n, device=dev)
        res = torch.randn(m, n, device=dev).to(BF)
        out = torch.empty(m, n, device=dev, dtype=BF)
        line = []
        for name, var in VARIANTS:
            for r in (None, res):
                i = [0]

                def f():
                    i[0] += 1
                    return ops.linear(xs[i[0] % 3], w, bias, residual=r, variant=var, out=out)
                try:
                    t = timeit(f)
                except RuntimeError:
                    t = float('nan')
                line.append(t)
                rows.setdefault((name, r is not None), []).append((k / 64, t))
        print(f"{k:6d} " + " ".join(f"{a:9.1f} {b:9.1f}" for a, b in zip(line[0::2], line[1::2])), flush=True)
    for (name, r), pts in rows.items():
        pts = [(x, y) for x, y in pts if y == y]
        if len(pts) < 3:
            continue
        mx = sum(x for x, _ in pts) / len(pts); my = sum(y for _, y in pts) / len(pts)
        b = sum((x - mx) * (y - my) for x, y in pts) / sum((x - mx) ** 2 for x, _ in pts)
        print(f"   {name:6s}{' +res' if r else '     '}: a = {my - b * mx:6.1f} us, b = {b:5.2f} us per 64 of K")

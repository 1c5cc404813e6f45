"""512x704 bucket, 64x88 level (M = 16 x 5632 = 90 112 rows = 352 row tiles of 256: 1.375 rounds of the 256 CUs, i.e. two rounds
at 69 %): does splitting the CFG batch into two launches whose tile counts fill whole rounds pay?  3x3 conv 320 -> 320 (K = 2 880)
through ops.conv on image sub-batches: all 16 | 11 + 5 (the 5 on two K slices: 242 + 220 workgroups) | 8 + 8 | 12 + 4 ...
usage: python tools/msplit_704.py"""
import math, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401,E402
from saspa_aug_amd import ops, weights as W  # noqa: E402
dev = torch.device('cuda:0'); BF = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(4):
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


for (b, h, w_, cin, cout) in ((16, 64, 88, 320, 320), (16, 64, 88, 640, 320), (16, 64, 64, 320, 320), (16, 64, 96, 320, 320)):
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(b, h, w_, cin, generator=g).to(dev, BF) for _ in range(3)]
    wt = W.to_chunk_major(torch.randn(cout, 9 * cin, generator=g) / math.sqrt(9 * cin), 9, BF).to(dev, BF)
    wt.saspa_korder = 1
    bias = torch.randn(cout, generator=g).to(dev)
    rv = torch.randn(b, cout, generator=g).to(dev)
    out = torch.empty(b, h, w_, cout, device=dev, dtype=BF)
    ref = ops.conv(xs[0], wt, bias, kh=3, kw=3, pad=1, rowvec=rv).clone()
    tiles_img = h * w_ / 256
    print(f"conv3x3 {b}x{h}x{w_}x{cin} -> {cout}: {b * tiles_img:.0f} row tiles ({tiles_img:.1f} per image)")
    i = [0]

    def whole():
        i[0] += 1
        return ops.conv(xs[i[0] % 3], wt, bias, kh=3, kw=3, pad=1, rowvec=rv, out=out)
    t0 = timeit(whole)
    print(f"   one launch (AUTO):                {t0:7.1f} us")
    for parts in ([(11, None), (5, 2)], [(11, None), (5, 3)], [(8, None), (8, None)], [(12, None), (4, 2)], [(11, None), (5, None)],
                  [(10, None), (6, 2)], [(11, 1), (5, 2)], [(9, None), (7, 2)]):
        if sum(p[0] for p in parts) != b:
            continue

        def split():
            i[0] += 1
            x = xs[i[0] % 3]
            s = 0
            for nb, ks in parts:
                ops.conv(x[s:s + nb], wt, bias, kh=3, kw=3, pad=1, rowvec=rv[s:s + nb], out=out[s:s + nb], ksplit=ks,
                         variant=ops.GEMM_WIDE if ks else ops.GEMM_AUTO)
                s += nb
            return out
        try:
            t = timeit(split)
            same = torch.equal(split(), ops.conv(xs[i[0] % 3], wt, bias, kh=3, kw=3, pad=1, rowvec=rv)) if all(k in (None, 1) for _, k in parts) else None
            print(f"   {str(parts):34s} {t:7.1f} us  ({t0 / t:.2f}x){'' if same is None else '  bit-identical to one launch: ' + str(same)}", flush=True)
        except RuntimeError as e:
            print(f"   {parts}: {str(e)[:80]}")

"""saspa_conv3x3_halo (GroupNorm + SiLU applied to the conv's input tile in LDS) against the launches it replaces, on the resnet
conv shapes of an SD-1.5 evaluation at 512x512 (CFG batch 16): per shape, microseconds of
    gn+conv : saspa_groupnorm_apply (epilogue statistics) + saspa_gemm (AUTO dispatch)      -- the round-4 path
    conv    : saspa_gemm alone on an already normalised input
    halo    : saspa_conv3x3_halo without a GroupNorm (plain halo conv)
    halo+gn : saspa_conv3x3_halo with the GroupNorm fused
HIP events around batches of back-to-back launches, inputs rotated over 4 buffers.  `twin`: launched with the sharing hint
(as inside the paired encoder region); every column then runs two copies on two streams and reports the pair time.
usage: python tools/halo_bench.py [all]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401,E402
from saspa_aug_amd import ops, weights as W  # noqa: E402

dev = torch.device('cuda:0')
BF = torch.bfloat16
SILU = ops.ACT_SILU


def timeit(fn, n=20):
    for _ in range(4):
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


#        name                      b   h   w   c0    c1    n    epilogue
SHAPES = [
    ("L0 conv1/2 320->320",       16, 64, 64, 320,  0,    320, "temb"),
    ("L0 conv2 res",              16, 64, 64, 320,  0,    320, "res"),
    ("L0 up conv1 640->320",      16, 64, 64, 320,  320,  320, "temb"),
    ("L0 up conv1 960->320",      16, 64, 64, 640,  320,  320, "temb"),
    ("L1 conv1 320->640",         16, 32, 32, 320,  0,    640, "temb"),
    ("L1 conv 640->640",          16, 32, 32, 640,  0,    640, "res"),
    ("L1 up conv1 1280->640",     16, 32, 32, 640,  640,  640, "temb"),
    ("L1 up conv1 1920->640",     16, 32, 32, 1280, 640,  640, "temb"),
    ("L2 conv1 640->1280",        16, 16, 16, 640,  0,    1280, "temb"),
    ("L2 conv 1280->1280",        16, 16, 16, 1280, 0,    1280, "res"),
    ("L2 up conv1 2560->1280",    16, 16, 16, 1280, 1280, 1280, "temb"),
]
if len(sys.argv) > 1 and sys.argv[1] == "l0":
    SHAPES = SHAPES[:4]

print(f"{'shape':26s} {'M':>6s} {'N':>5s} {'K':>6s} | {'gn+conv':>9s} {'(gn':>7s} {'conv)':>7s} | {'halo':>7s} {'halo+gn':>8s} | {'TF/s halo+gn':>12s} {'speedup':>8s}")
for name, b, h, w_, c0, c1, n, epi in SHAPES:
    g = torch.Generator().manual_seed(1)
    ctot = c0 + c1
    m = b * h * w_
    NB = 4
    xs = [torch.randn(b, h, w_, c0, generator=g).to(dev, BF) for _ in range(NB)]
    x2s = [torch.randn(b, h, w_, c1, generator=g).to(dev, BF) for _ in range(NB)] if c1 else [None] * NB
    pk = torch.randn(n, 9 * ctot, generator=g) / math.sqrt(9 * ctot)
    w64 = W.to_chunk_major(pk, 9, BF).to(dev, BF)
    w64.saspa_korder = 1
    w32 = W.to_chunk32_major(pk).to(dev, BF)
    gamma = (1 + 0.1 * torch.randn(ctot, generator=g)).to(dev)
    beta = (0.1 * torch.randn(ctot, generator=g)).to(dev)
    gb32 = W.pack_gamma_beta32(gamma.cpu(), beta.cpu()).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    rv = torch.randn(b, n, generator=g).to(dev) if epi == "temb" else None
    res = torch.randn(b, h, w_, n, generator=g).to(dev, BF) if epi == "res" else None
    out = torch.empty(b, h, w_, n, device=dev, dtype=BF)
    # epilogue statistics of the inputs, as their producers leave them (unit 10)
    def with_stats(t):
        if t is None:
            return None
        c = t.shape[-1]
        v = t.float().reshape(-1, 128, c // 10, 10)
        st = torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1).contiguous()
        t.saspa_gn = (st, 10, t.data_ptr(), t._version)      # what a producing conv's epilogue leaves (ops._gn_stats_for)
        return t
    xs = [with_stats(t) for t in xs]
    x2s = [with_stats(t) for t in x2s]
    hn = torch.empty(b, h, w_, ctot, device=dev, dtype=BF)
    k = [0]

    def nxt():
        k[0] = (k[0] + 1) % NB
        return xs[k[0]], x2s[k[0]]

    def f_gn():
        x, x2 = nxt()
        ops.groupnorm(x, gamma, beta, 32, 1e-5, SILU, x2=x2, out=hn)

    def f_conv():
        ops.conv(hn, w64, bias, kh=3, kw=3, pad=1, rowvec=rv, residual=res, out=out, gn_unit=10)

    def f_two():
        f_gn()
        f_conv()

    def f_halo():
        x, x2 = nxt()
        assert ops.conv_gn(x, None, w32, bias, x2=x2, rowvec=rv, residual=res, out=out, gn_unit=10) is not None

    def f_halo_gn():
        x, x2 = nxt()
        assert ops.conv_gn(x, (gb32, 32, 1e-5, SILU), w32, bias, x2=x2, rowvec=rv, residual=res, out=out, gn_unit=10) is not None

    t_two, t_gn, t_conv, t_h, t_hg = timeit(f_two), timeit(f_gn), timeit(f_conv), timeit(f_halo), timeit(f_halo_gn)
    tf = 2.0 * m * n * 9 * ctot / (t_hg * 1e-6) / 1e12
    print(f"{name:26s} {m:6d} {n:5d} {9 * ctot:6d} | {t_two:9.1f} {t_gn:7.1f} {t_conv:7.1f} | {t_h:7.1f} {t_hg:8.1f} | {tf:12.0f} {t_two / t_hg:8.2f}", flush=True)
    del xs, x2s, w64, w32

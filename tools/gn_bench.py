"""GroupNorm(+SiLU) apply pass at the shapes of a batch-8 512x512 step: time and HBM bytes per second (read + write of the
tensor), with the statistics taken from a producer conv's epilogue (one launch) and with the two-launch form.
NOTE: launched eagerly from Python, one call per iteration: below ~20 us per call the figure is the host's launch rate, not
the kernel (the 16x16 / 8x8 rows); read those shapes from the rocprofv3 kernel trace instead.
usage: python tools/gn_bench.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
dev = torch.device('cuda:0'); BF = torch.bfloat16
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (b, h, w, c, c2) in ((16, 64, 64, 320, 0), (16, 64, 64, 320, 320), (16, 64, 64, 640, 320), (16, 32, 32, 640, 0), (16, 32, 32, 640, 640),
                         (16, 16, 16, 1280, 0), (16, 16, 16, 1280, 1280), (16, 8, 8, 1280, 1280)):
    ct = c + c2
    gamma, beta = torch.randn(ct, device=dev), torch.randn(ct, device=dev)
    # producers: 1x1 convs that leave epilogue statistics (gn_unit = 10)
    def producer(cc):
        xin = torch.randn(b, h, w, cc, device=dev).to(BF)
        if cc == 320:       # (the 1x1 layer of this size runs on the A-stationary kernel, which leaves no statistics: a 3x3 conv as in a ResNet block)
            wt = (torch.randn(cc, 9 * cc, device=dev) / math.sqrt(9 * cc)).to(BF)
            return ops.conv(xin, wt, kh=3, kw=3, pad=1, gn_unit=10)
        wt = (torch.randn(cc, cc, device=dev) / math.sqrt(cc)).to(BF)
        return ops.conv(xin, wt, gn_unit=10)
    xs = [producer(c) for _ in range(3)]
    x2s = [producer(c2) for _ in range(3)] if c2 else [None] * 3
    outs = [torch.empty(b, h, w, ct, device=dev, dtype=BF) for _ in range(3)]
    i = [0]
    def fused():
        j = i[0] % 3; i[0] += 1
        ops.groupnorm(xs[j], gamma, beta, 32, 1e-5, ops.ACT_SILU, x2=x2s[j], out=outs[j])
    has = hasattr(xs[0], "saspa_gn")
    tf = timeit(fused)
    os.environ["SASPA_GN_FUSE"] = "0"
    t2 = timeit(fused)
    os.environ["SASPA_GN_FUSE"] = "1"
    mb = 2 * b * h * w * ct * 2 / 1e6
    print(f"GN+SiLU [{b},{h},{w},{c}{'+' + str(c2) if c2 else ''}] {mb:6.1f} MB r+w: epilogue statistics{'' if has else ' (n/a)'} {tf:6.1f} us = {mb / tf:5.2f} TB/s | "
          f"statistics pass + apply {t2:6.1f} us", flush=True)

"""fp8 W8A8 (LayerNorm-quantise + saspa_gemm_fp8) against the bf16 path (LayerNorm + saspa_gemm) on the SDXL transformer
shapes at 1024x1024, batch 8 (HIP events)."""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops, weights as W
dev = torch.device('cuda:0'); BF = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (m, c) in [(32768, 640), (8192, 1280), (8192, 640), (2048, 1280)]:
    xs = [torch.randn(m, c, device=dev).to(BF) for _ in range(3)]
    g, be = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    i = [0]
    def x():
        i[0] += 1
        return xs[i[0] % 3]
    for name, n, geglu in (("q", c, False), ("geglu", 8 * c, True)):
        w = torch.randn(n, c) / math.sqrt(c); b = torch.randn(n)
        if geglu:
            w16, b16 = W.pack_geglu(w, b)
            w8, b8 = W.pack_geglu_tile(w, b, 128)
        else:
            w16, b16, w8, b8 = w, b, w, b
        wq, sw = W.quantize_fp8(w8)
        w16d, b16d, wqd, swd, b8d = w16.to(dev, BF), b16.to(dev), wq.to(dev), sw.to(dev), b8.to(dev)
        act = ops.ACT_GEGLU if geglu else ops.ACT_NONE
        ln16 = timeit(lambda: ops.layernorm(x(), g, be))
        ln8 = timeit(lambda: ops.layernorm_quant_fp8(x(), g, be))
        n1 = ops.layernorm(xs[0], g, be); q8, s8 = ops.layernorm_quant_fp8(xs[0], g, be)
        t16 = timeit(lambda: ops.linear(n1, w16d, b16d, act=act))
        t8 = timeit(lambda: ops.linear_fp8(q8, s8, wqd, swd, b8d, act=act))
        fl = 2.0 * m * n * c
        print(f"M={m} C={c} {name:6s} N={n}: LN bf16 {ln16:6.1f} us / quant {ln8:6.1f} us | GEMM bf16 {t16:7.1f} us ({fl / t16 / 1e6:6.0f} TF) / fp8 {t8:7.1f} us ({fl / t8 / 1e6:6.0f} TF)", flush=True)

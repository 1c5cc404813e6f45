"""saspa_xattn_block against the three launches it replaces, at the level-0 shape of a batch-8 512x512 step (M = 65 536 tokens,
16 samples of 4 096, 77 keys).  usage: python tools/xattn_bench.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops, weights as W
from saspa_aug_amd.models import ATTN_LOG2E, attention_core
dev = torch.device('cuda:0'); BF = torch.bfloat16
C, HEADS, D, NK = 320, 8, 40, 77
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (nsamp, ntok) in ((16, 4096), (16, 5632)):
    m = nsamp * ntok
    xs = [torch.randn(nsamp, ntok, C, device=dev).to(BF) for _ in range(3)]
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    wq = torch.randn(C, C) / math.sqrt(C) * (D ** -0.5 * ATTN_LOG2E); wo = torch.randn(C, C) / math.sqrt(C); bo = torch.randn(C)
    k = torch.randn(nsamp, NK, C, device=dev).to(BF); v = torch.randn(nsamp, NK, C, device=dev).to(BF)
    w, bias = W.pack_xattn_w(wq, wo, bo); w, bias = w.to(dev, BF), bias.to(dev)
    kf, vf = W.xattn_kv_fragments(k, v)
    vt = torch.zeros((nsamp, C, 80), device=dev, dtype=BF); vt[:, :, :NK] = v.transpose(1, 2)
    wqd, wod, bod = wq.to(dev, BF), wo.to(dev, BF), bo.to(dev)
    i = [0]
    def fused():
        x = xs[i[0] % 3]; i[0] += 1
        return ops.xattn_block(x, (gamma, beta, 1e-5), w, bias, kf, vf, NK, ntok)
    def chain():
        x = xs[i[0] % 3]; i[0] += 1
        if ops.linear_ln_fusable(x, wqd):
            q = ops.linear(x, wqd, ln=(gamma, beta, 1e-5))
        else:
            q = ops.linear(ops.layernorm(x, gamma, beta), wqd)
        o = attention_core(q, k, vt, HEADS, ntok, NK, prescaled=True)
        return ops.linear(o, wod, bod, residual=x)
    tf, tc = timeit(fused), timeit(chain)
    fl = 4.0 * m * C * C + 4.0 * m * NK * C
    print(f"M={m} ({nsamp} x {ntok}), {NK} keys: fused {tf:7.1f} us ({fl / tf / 1e6:6.0f} TF/s, {3 * m * C * 2 / tf / 1e6:5.0f} GB/s of x / residual / y) | three launches {tc:7.1f} us", flush=True)

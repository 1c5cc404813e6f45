"""A-stationary K = 320 GEMM (saspa_gemm_as.hip) against the tiled / wave-specialised / wide kernels on the level-0 pointwise
shapes (M = 65 536 tokens of a 512x512 batch-8 CFG evaluation, 90 112 at 512x704), and the fused forms against the launches
they replace (LayerNorm + GEMM; LayerNorm + Q|K GEMM + batched V^T GEMM).  HIP events, 20 launches each."""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import models, ops, weights as W
dev = torch.device('cuda:0'); BF = torch.bfloat16
K = 320
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
names = {ops.GEMM_AS: "as", ops.GEMM_TILED: "tiled", ops.GEMM_WS: "ws", ops.GEMM_WIDE: "wide"}
for m in (65536, 90112):
    xs = [torch.randn(m, K, device=dev).to(BF) for _ in range(3)]
    res = torch.randn(m, 320, device=dev).to(BF)
    gamma, beta = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
    for (n, act, residual) in ((320, ops.ACT_NONE, False), (320, ops.ACT_NONE, True), (640, ops.ACT_NONE, False), (2560, ops.ACT_GEGLU, False)):
        w32, b32 = torch.randn(n, K) / math.sqrt(K), torch.randn(n)
        if act == ops.ACT_GEGLU:
            w32, b32 = W.pack_geglu(w32, b32)
        w, b = w32.to(dev, BF), b32.to(dev)
        row, i = [], [0]
        for v in (ops.GEMM_AS, ops.GEMM_TILED, ops.GEMM_WS, ops.GEMM_WIDE):
            def f():
                i[0] += 1
                ops.linear(xs[i[0] % 3], w, b, act=act, residual=res if residual else None, variant=v)
            try:
                us = timeit(f)
                row.append(f"{names[v]} {us:6.1f}us {2.0 * m * n * K / us / 1e6:5.0f}TF")
            except RuntimeError:
                row.append(f"{names[v]} n/a")
        def f_ln():
            i[0] += 1
            ops.linear(xs[i[0] % 3], w, b, act=act, ln=(gamma, beta, 1e-5))
        def f_two():
            i[0] += 1
            ops.linear(ops.layernorm(xs[i[0] % 3], gamma, beta, 1e-5), w, b, act=act, variant=ops.GEMM_TILED if act == ops.ACT_NONE else ops.GEMM_WS)
        if not residual:
            row.append(f"LN fused {timeit(f_ln):6.1f}us vs LN + tiled/ws {timeit(f_two):6.1f}us")
        print(f"M={m} N={n} act={act} res={int(residual)}: " + " | ".join(row), flush=True)
    # Q | K | V^T
    b_, ntok = 16, m // 16
    h = torch.randn(b_, ntok, K, device=dev).to(BF)
    wqk = (torch.randn(640, K) / math.sqrt(K)).to(dev, BF); wv = (torch.randn(320, K) / math.sqrt(K)).to(dev, BF)
    wall = torch.cat([wqk, wv], 0).contiguous()
    vt = torch.empty((b_, 320, ntok), device=dev, dtype=BF)
    def f_one():
        ops.linear(h, wall, None, ln=(gamma, beta, 1e-5), out_t=vt, n_split=640, rows_per_batch=ntok)
    def f_three():
        n1 = ops.layernorm(h, gamma, beta, 1e-5)
        ops.linear(n1, wqk, variant=ops.GEMM_TILED)
        models.project_vt(n1, wv, ntok)
    print(f"M={m} LN + Q|K + V^T: one launch {timeit(f_one):6.1f}us vs three {timeit(f_three):6.1f}us", flush=True)

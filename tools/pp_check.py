"""A/B of the GEMM kernel variants on the SD-1.5 layer shapes: time + max error vs a torch fp32 conv.
usage: python tools/pp_check.py            (parent: one child per SASPA_GEMM_PP mode)
"""
import os, sys, math, subprocess

# (batch, H, W, Cin, Cout, k, upsample, ksplit_override)
CONVS = [
    (16, 64, 64, 320, 320, 3, 0), (16, 64, 64, 640, 320, 3, 0), (16, 64, 64, 960, 320, 3, 0),
    (16, 32, 32, 640, 640, 3, 0), (16, 32, 32, 1280, 640, 3, 0), (16, 32, 32, 320, 640, 3, 0),
    (16, 16, 16, 1280, 1280, 3, 0), (16, 16, 16, 2560, 1280, 3, 0), (16, 8, 8, 1280, 1280, 3, 0),
    (8, 128, 128, 512, 512, 3, 0), (8, 256, 256, 256, 256, 3, 0), (8, 512, 512, 128, 128, 3, 0),
    (8, 128, 128, 512, 512, 3, 1), (8, 256, 256, 256, 256, 3, 1), (8, 64, 64, 512, 512, 3, 0),
    (16, 64, 64, 1280, 320, 1, 0), (16, 32, 32, 2560, 640, 1, 0), (16, 16, 16, 5120, 1280, 1, 0),
    (16, 16, 16, 1280, 1280, 1, 0), (16, 64, 64, 320, 320, 1, 0),
]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
    import saspa_aug_amd  # noqa: F401
    from saspa_aug_amd import ops, weights
    dev = torch.device('cuda:0')
    check = os.environ.get("PP_CHECK_ERR", "1") == "1"

    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    for (b, h, w_, ci, co, k, up) in CONVS:
        g = torch.Generator(device='cpu').manual_seed(ci * 7 + co + h)
        nbuf = max(2, min(8, int(400e6 // (b * h * w_ * ci * 2))))
        xs = [torch.randn(b, h, w_, ci, generator=g).bfloat16().to(dev) for _ in range(nbuf)]
        w4 = (torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)).bfloat16()
        bias = torch.randn(co, generator=g).float().to(dev)
        wp = weights.pack_conv(w4).to(dev)
        outs = [None] * nbuf
        i = [0]
        def f():
            j = i[0] % nbuf; i[0] += 1
            outs[j] = ops.conv(xs[j], wp, bias, kh=k, kw=k, pad=k // 2, upsample=bool(up), out=outs[j])
        us = timeit(f)
        ho, wo = (2 * h, 2 * w_) if up else (h, w_)
        flops = 2.0 * b * ho * wo * co * ci * k * k
        err = float('nan')
        if check:
            j = 0
            outs[j] = ops.conv(xs[j], wp, bias, kh=k, kw=k, pad=k // 2, upsample=bool(up), out=outs[j])
            xin = xs[j][:2].permute(0, 3, 1, 2).float()
            if up: xin = F.interpolate(xin, scale_factor=2, mode='nearest')
            ref = F.conv2d(xin, w4.to(dev).float(), bias, padding=k // 2).permute(0, 2, 3, 1)
            err = (outs[j][:2].float() - ref).abs().max().item()
            # last images too (tile tails)
            xin = xs[j][-1:].permute(0, 3, 1, 2).float()
            if up: xin = F.interpolate(xin, scale_factor=2, mode='nearest')
            ref = F.conv2d(xin, w4.to(dev).float(), bias, padding=k // 2).permute(0, 2, 3, 1)
            err = max(err, (outs[j][-1:].float() - ref).abs().max().item())
        print(f"{b}x{h}x{w_} {ci}->{co} k{k} up{up} | {us:8.1f} us {flops / us * 1e-6:7.1f} TF/s err {err:.4f}", flush=True)
else:
    modes = sys.argv[1:] or ["0", "4"]
    res = {}
    for mode in modes:
        env = dict(os.environ, SASPA_GEMM_PP=mode)
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        lines = [l for l in out.stdout.strip().splitlines() if "|" in l]
        if not lines:
            print(mode, "FAILED", out.stderr[-2000:])
        res[mode] = lines
    n = max(len(v) for v in res.values())
    for i in range(n):
        name = None
        cols = []
        for mode in modes:
            if i < len(res[mode]):
                name, rest = res[mode][i].split("|")
                cols.append(f"[{mode}]{rest}")
        print(name, " ".join(cols))

"""K-loop ablation of the 8-wave ping-pong GEMM (diagnostic; needs `make ABLATION=1`)."""
import os as _os; _os.environ.setdefault("SASPA_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "saspa-aug_amd", "libsaspa_hip_abl.so"))
import os, sys, math, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
    import saspa_aug_amd  # noqa: F401
    from saspa_aug_amd import ops
    dev = torch.device('cuda:0')
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    out = []
    x = torch.randn(16, 64, 64, 640, device=dev).bfloat16()
    w = (torch.randn(320, 5760, device=dev) / 70).bfloat16()
    out.append(timeit(lambda: ops.conv(x, w, kh=3, kw=3, pad=1)))
    x1 = torch.randn(8, 128, 128, 512, device=dev).bfloat16()
    w1 = (torch.randn(512, 4608, device=dev) / 70).bfloat16()
    out.append(timeit(lambda: ops.conv(x1, w1, kh=3, kw=3, pad=1)))
    x3 = torch.randn(65536, 1280, device=dev).bfloat16(); w3 = (torch.randn(1280, 1280, device=dev) / 36).bfloat16()
    o3 = torch.empty(65536, 1280, device=dev, dtype=torch.bfloat16)
    out.append(timeit(lambda: ops.linear(x3, w3, out=o3)))
    print(" ".join(f"{v:9.1f}" for v in out))
else:
    pp = sys.argv[1] if len(sys.argv) > 1 else "5"
    loop = sys.argv[2] if len(sys.argv) > 2 else "1"     # SASPA_GEMM_PP_LOOP: 0 ping-pong, 1 asymmetric one-barrier loop
    print("SASPA_GEMM_PP_LOOP =", loop)
    print(f"PP={pp} variant                          conv 65536x320x5760   conv 131072x512x4608   lin 65536x1280x1280  (us)")
    for name, abl in (("full", 0), ("no MFMA", 1), ("no LDS reads (MFMA on stale regs)", 2), ("no DMA", 4), ("no barriers", 8),
                      ("DMA + barriers only", 3), ("MFMA + barriers only", 6), ("reads + barriers only", 5), ("barriers only", 7),
                      ("MFMA only", 14)):
        env = dict(os.environ, SASPA_GEMM_ABLATE=str(abl), SASPA_GEMM_PP=pp, SASPA_GEMM_PP_LOOP=loop)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print(f"{name:40s} {r}")

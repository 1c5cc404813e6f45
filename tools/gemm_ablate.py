"""K-loop ablation of the LDS-DMA GEMM (diagnostic): time full / no-MFMA / DMA-only / no-DMA builds."""
import os as _os; _os.environ.setdefault("SASPA_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "saspa-aug_amd", "libsaspa_hip_abl.so"))
import os, sys, math, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
    import saspa_aug_amd
    from saspa_aug_amd import ops
    from saspa_aug_amd import weights as W
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
    x = torch.randn(16, 64, 64, 320, device=dev).bfloat16()
    w = (torch.randn(320, 2880, device=dev) / 50).bfloat16()
    out.append(timeit(lambda: ops.conv(x, w, kh=3, kw=3, pad=1)))
    x2 = torch.randn(65536, 320, device=dev).bfloat16(); w2 = (torch.randn(320, 320, device=dev) / 18).bfloat16()
    o2 = torch.empty(65536, 320, device=dev, dtype=torch.bfloat16)
    out.append(timeit(lambda: ops.linear(x2, w2, out=o2)))
    x3 = torch.randn(65536, 1280, device=dev).bfloat16(); w3 = (torch.randn(1280, 1280, device=dev) / 36).bfloat16()
    o3 = torch.empty(65536, 1280, device=dev, dtype=torch.bfloat16)
    out.append(timeit(lambda: ops.linear(x3, w3, out=o3)))
    print(" ".join(f"{v:9.1f}" for v in out))
else:
    print("variant            conv 65536x320x2880   lin 65536x320x320   lin 65536x1280x1280  (us)")
    for name, abl in (("full", 0), ("no MFMA", 1), ("DMA+barriers only", 2), ("no DMA", 4), ("no DMA, no MFMA", 5), ("barriers only", 6), ("DMA + MFMA on stale regs (no LDS reads)", 8), ("MFMA on stale regs only", 12)):
        env = dict(os.environ, SASPA_GEMM_ABLATE=str(abl))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print(f"{name:40s} {r}")

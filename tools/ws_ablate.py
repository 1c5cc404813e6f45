"""Role ablation of the wave-specialised GEMM (needs `make ABLATION=1`): which wave group bounds a K-tile step."""
import os as _os; _os.environ.setdefault("SASPA_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "saspa-aug_amd", "libsaspa_hip_abl.so"))
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    import saspa_aug_amd  # noqa: F401
    from saspa_aug_amd import ops, weights as W
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
    for (m, k, f) in ((65536, 320, 1280),):
        x = torch.randn(m, k, device=dev).bfloat16()
        w = (torch.randn(2 * f, k) / k ** 0.5); b = torch.randn(2 * f)
        wp, bp = W.pack_geglu(w, b)
        wp, bp = wp.to(dev, torch.bfloat16), bp.to(dev)
        o = torch.empty(m, f, device=dev, dtype=torch.bfloat16)
        out.append(timeit(lambda: ops.linear(x, wp, bp, act=ops.ACT_GEGLU, out=o, variant=3)))
    for (m, k, n) in ((65536, 320, 2560), (65536, 320, 320), (16384, 640, 640), (65536, 1280, 320)):
        x = torch.randn(m, k, device=dev).bfloat16(); w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
        o = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        out.append(timeit(lambda: ops.linear(x, w, out=o, variant=3)))
    print(" ".join(f"{v:9.1f}" for v in out))
else:
    print("WS variant                  geglu 65536x2560x320 | plain 65536x2560x320  65536x320x320  16384x640x640  65536x320x1280 (us)")
    for name, abl in (("full", 0), ("no MFMA/reads", 1), ("no DMA", 2), ("no epilogue", 4), ("no DMA, no epilogue", 6), ("no MFMA, no epilogue", 5), ("barriers only", 7)):
        env = dict(os.environ, SASPA_GEMM_ABLATE=str(abl))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(f"{name:26s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}")

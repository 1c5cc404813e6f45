"""Role ablation of the A-stationary K = 320 GEMM (diagnostic): which of W fragment reads / MFMAs / epilogue / DMA / barrier
bounds a 64-column slice.  The ablation is a COMPILE-TIME constant (-DSASPA_AS_ABLATE=bits; a run-time switch inside the loop
changes what it measures): `python tools/as_ablate.py build` (needs hipcc; run where the repo was built) makes one library per
variant next to libsaspa_hip.so, `python tools/as_ablate.py` (GPU box) times them, one child process per library."""
import math, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(HERE, "..", "saspa-aug_amd")
VARIANTS = (("full", 0), ("no W fragment reads", 1), ("no MFMA", 2), ("no epilogue", 4), ("no DMA", 8), ("no barrier", 16),
            ("DMA + barriers only", 7), ("reads + MFMA (no epilogue, no DMA)", 12), ("epilogue only", 11),
            ("nothing but the A load + barriers", 15), ("stores out of range (issued, no bytes move)", 32))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    cs = os.path.join(PKG, "csrc")
    objs = [os.path.join(cs, f) for f in sorted(os.listdir(cs)) if f.endswith(".o") and not f.endswith(".abl.o") and f != "saspa_gemm_as.o"]
    for _, abl in VARIANTS:
        o = f"/tmp/saspa_gemm_as_{abl}.o"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{os.path.join(HERE, '..', 'include')}",
                               f"-DSASPA_AS_ABLATE={abl}", "-c", os.path.join(cs, "saspa_gemm_as.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", *objs, o, "-o", os.path.join(PKG, f"libsaspa_hip_as{abl}.so")])
        print("built", abl, flush=True)
elif len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.join(HERE, ".."))
    import saspa_aug_amd  # noqa: F401
    from saspa_aug_amd import ops, weights as W
    dev = torch.device("cuda:0"); BF = torch.bfloat16
    K, m = 320, 65536
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    x = torch.randn(m, K, device=dev).to(BF)
    row = []
    for (n, act) in ((640, ops.ACT_NONE), (2560, ops.ACT_GEGLU)):
        w32, b32 = torch.randn(n, K) / math.sqrt(K), torch.randn(n)
        if act == ops.ACT_GEGLU:
            w32, b32 = W.pack_geglu(w32, b32)
        w, b = w32.to(dev, BF), b32.to(dev)
        row.append(timeit(lambda: ops.linear(x, w, b, act=act, variant=ops.GEMM_AS)))
    print("  ".join(f"{v:11.1f}" for v in row))
else:
    print(f"{'variant':44s} N=640 plain  N=2560 GEGLU   (us at M = 65 536; 10 / 40 slices)")
    for name, abl in VARIANTS:
        env = dict(os.environ, SASPA_HIP_LIB=os.path.join(PKG, f"libsaspa_hip_as{abl}.so"))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(f"{name:44s} {(r.stdout.strip().splitlines() or [r.stderr[-200:]])[-1]}", flush=True)

"""Where a step of the A-stationary GEMM spends its time: s_memtime stamps of workgroup 0's waves 0 and 7 (top of step | after
the DMA wait | after the barrier | after the DMA issue; the rest of the step is MFMAs + the previous slice's epilogue).
`python tools/as_stamps.py build` (needs hipcc) makes libsaspa_hip_asst.so with -DSASPA_AS_STAMPS; `python tools/as_stamps.py`
runs it on the GPU box.  s_memtime ticks at 100 MHz on gfx950 (10 ns)."""
import math, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); PKG = os.path.join(HERE, "..", "saspa-aug_amd")
LIB = os.environ.get("SASPA_AS_STAMP_LIB") or os.path.join(PKG, "libsaspa_hip_asst.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    cs = os.path.join(PKG, "csrc")
    objs = [os.path.join(cs, f) for f in sorted(os.listdir(cs)) if f.endswith(".o") and not f.endswith(".abl.o") and f != "saspa_gemm_as.o"]
    o = "/tmp/saspa_gemm_as_st.o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{os.path.join(HERE, '..', 'include')}",
                           "-DSASPA_AS_STAMPS", *sys.argv[2:], "-c", os.path.join(cs, "saspa_gemm_as.hip"), "-o", o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", *objs, o, "-o", LIB])
    print("built", LIB)
    sys.exit(0)
os.environ["SASPA_HIP_LIB"] = LIB
import ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.join(HERE, ".."))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import _lib, ops, weights as W
dev = torch.device("cuda:0"); BF = torch.bfloat16
K, m = 320, 65536
x = torch.randn(m, K, device=dev).to(BF)
for (n, act) in ((640, ops.ACT_NONE), (2560, ops.ACT_GEGLU)):
    w32, b32 = torch.randn(n, K) / math.sqrt(K), torch.randn(n)
    if act == ops.ACT_GEGLU:
        w32, b32 = W.pack_geglu(w32, b32)
    w, b = w32.to(dev, BF), b32.to(dev)
    nout = n // 2 if act == ops.ACT_GEGLU else n
    out = torch.empty(m, nout, device=dev, dtype=BF)
    dbg = torch.zeros(8192, device=dev, dtype=torch.int64)
    p = ops._linear_params(x, w, b, None, out, 1.0, act, None, ops.GEMM_AS, m, n, K)
    p.ksplit, p.workspace = 1, C.c_void_p(dbg.data_ptr())
    lib = _lib.load()
    for _ in range(3):
        _lib.check(lib.saspa_gemm(C.byref(p), ops._stream()), "saspa_gemm")
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    ns = n // 64
    for wv, name in ((0, "wave 0"), (1, "wave 7")):
        st = d[wv * 4096: wv * 4096 + ns * 4].reshape(ns, 4)[1:]          # steps 1 .. ns-1
        tick = 10.0     # ns per s_memtime tick
        wait = (st[:, 1] - st[:, 0]) * tick; bar = (st[:, 2] - st[:, 1]) * tick; dma = (st[:, 3] - st[:, 2]) * tick
        rest = (st[1:, 0] - st[:-1, 3]) * tick
        print(f"N={n} act={act} {name}: per step (ns, median over {ns - 1} steps)  DMA wait {np.median(wait):6.0f}  barrier {np.median(bar):6.0f}  "
              f"DMA issue {np.median(dma):6.0f}  MFMA + epilogue {np.median(rest):6.0f}  | step {np.median(st[1:, 0] - st[:-1, 0]) * tick:6.0f}", flush=True)

"""Cycle timeline of workgroup 0 of the wave-specialised GEMM (ablation build, SASPA_GEMM_ABLATE=8): s_memtime stamps of
MMA wave 0 (tags 1..5) and LE wave 4 (tags 11..15)."""
import os, sys
os.environ.setdefault("SASPA_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "saspa-aug_amd", "libsaspa_hip_abl.so"))
os.environ["SASPA_GEMM_ABLATE"] = os.environ.get("SASPA_GEMM_ABLATE", "8")
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops, weights as W
dev = torch.device('cuda:0')
keep = {}
orig = ops._set_splitk
def capture(p, m, n, k, t, force=None):
    ws = torch.zeros((4096,), device=t.device, dtype=torch.float32)
    import ctypes as C
    p.ksplit, p.workspace = 1, C.c_void_p(ws.data_ptr())
    keep["ws"] = ws
    return ws
ops._set_splitk = capture
NAMES = {1: "mma: arrive B(g)", 2: "mma: pass B(g)", 3: "mma: arrive Bx", 4: "mma: pass Bx", 5: "mma: dump done",
         11: "ld: arrive B(g)", 12: "ld: pass B(g)", 13: "ld: dma issued", 15: "ld: vmcnt passed"}
def run(m, n, k, geglu=False):
    x = torch.randn(m, k, device=dev).bfloat16()
    if geglu:
        w = torch.randn(n, k) / k ** 0.5; b = torch.randn(n)
        wp, bp = W.pack_geglu(w, b); wp, bp = wp.to(dev, torch.bfloat16), bp.to(dev)
        f = lambda: ops.linear(x, wp, bp, act=ops.ACT_GEGLU, variant=3)
    else:
        w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16(); b = torch.randn(n, device=dev)
        f = lambda: ops.linear(x, w, b, variant=3)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    raw = keep["ws"].cpu().view(torch.int64)
    for base, who in ((0, "MMA wave 0"), (512, "loader wave 4")):
        st = [(int(v) >> 8, int(v) & 255) for v in raw[base:base + 500].tolist() if v != 0]
        print(f"--- {who}: {m}x{n}x{k} geglu={geglu}  ({len(st)} stamps; cycles since the first) ---")
        t0 = st[0][0]
        prev = t0
        for t, tag in st[:75]:
            print(f"{t - t0:8d}  +{t - prev:6d}  {NAMES.get(tag, tag)}")
            prev = t
run(65536, 320, 320)

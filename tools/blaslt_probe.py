"""Do the library GEMMs (hipBLASLt behind torch.nn.functional.linear / torch.addmm) beat the hand-written pointwise kernels on the
mid-size shapes of a step?  Same process, HIP events, interleaved; bias + residual in both arms.  A probe for a dispatch decision:
the product path calls neither torch GEMMs nor hipBLASLt today.  usage: python tools/blaslt_probe.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
dev = torch.device('cuda:0'); BF = torch.bfloat16
SHAPES = [(16384, 640, 640, True), (16384, 640, 2560, True), (16384, 1920, 640, False), (16384, 640, 640, False),
          (4096, 1280, 1280, True), (4096, 1280, 5120, True), (4096, 3840, 1280, False), (4096, 1280, 1280, False),
          (65536, 320, 320, True), (65536, 320, 1280, True), (65536, 960, 320, False),
          (2056, 1024, 4096, True), (2056, 4096, 1024, False), (2056, 1024, 1024, True), (1024, 1280, 1280, True), (1024, 1280, 5120, True)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (m, n, k, res) in SHAPES:
    x = [torch.randn(m, k, device=dev).to(BF) for _ in range(3)]
    w = (torch.randn(n, k, device=dev) / math.sqrt(k)).to(BF)
    b = torch.randn(n, device=dev)
    bb = b.to(BF)
    r = [torch.randn(m, n, device=dev).to(BF) for _ in range(3)]
    out = [torch.empty(m, n, device=dev, dtype=BF) for _ in range(3)]
    i = [0]
    def mine():
        j = i[0] % 3; i[0] += 1
        ops.linear(x[j], w, b, residual=r[j] if res else None, out=out[j])
    def lib():
        j = i[0] % 3; i[0] += 1
        if res:
            y = torch.nn.functional.linear(x[j], w, bb)
            torch.add(y, r[j], out=out[j])
        else:
            torch.nn.functional.linear(x[j], w, bb)
    def lib_gemm_only():
        j = i[0] % 3; i[0] += 1
        torch.mm(x[j], w.t(), out=out[j])
    ts = {"mine": [], "lib": [], "mm": []}
    for rnd in range(4):
        for name, fn in (("mine", mine), ("lib", lib), ("mm", lib_gemm_only)):
            t = timeit(fn)
            if rnd: ts[name].append(t)
    med = {k2: sorted(v)[len(v) // 2] for k2, v in ts.items()}
    fl = 2.0 * m * n * k
    print(f"({m}, {n}, {k}){' +res' if res else '     '}: saspa_gemm {med['mine']:6.1f} us ({fl / med['mine'] / 1e6:5.0f} TF/s) | torch linear(+add) {med['lib']:6.1f} us | "
          f"torch.mm alone {med['mm']:6.1f} us ({fl / med['mm'] / 1e6:5.0f} TF/s)", flush=True)

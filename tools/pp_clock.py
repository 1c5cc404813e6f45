"""Per-workgroup K-loop timeline and in-kernel clock of the ping-pong GEMM (diagnostic; needs `make ABLATION=1`)."""
import os as _os; _os.environ.setdefault("SASPA_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "saspa-aug_amd", "libsaspa_hip_abl.so"))
import os, sys, time
os.environ.setdefault("SASPA_GEMM_PP", "5"); os.environ["SASPA_GEMM_STAMP"] = "1"
import torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
dev = torch.device('cuda:0')
# usage: pp_clock.py [input channels = 640] [res: 1 = residual + GroupNorm-statistics epilogue]
CIN = int(sys.argv[1]) if len(sys.argv) > 1 else 640
RES = len(sys.argv) > 2 and sys.argv[2] == "1"
x = torch.randn(16, 64, 64, CIN, device=dev).bfloat16()
w = (torch.randn(320, 9 * CIN, device=dev) / 70).bfloat16()
big = torch.zeros(17, 64, 64, 320, device=dev, dtype=torch.bfloat16)
out = big[:16]
res = torch.randn(16, 64, 64, 320, device=dev).bfloat16() if RES else None
_conv = ops.conv
ops_conv = lambda *a, **k: _conv(*a, residual=res, gn_unit=(10 if RES else None), **k)
t0 = time.time()
while time.time() - t0 < 2.5:
    for _ in range(200): ops_conv(x, w, kh=3, kw=3, pad=1, out=out)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops_conv(x, w, kh=3, kw=3, pad=1, out=out); e1.record(); torch.cuda.synchronize()
st = big[16].reshape(-1).view(torch.int64)[:16 * 256].reshape(256, 16).cpu()
s0, s1, clk, end = st[:, 0], st[:, 1], st[:, 2] >> 20, st[:, 3]
pre = st[:, 2] & 0xfffff
base = s0.min()
f = lambda t: f"min {float(t.min()) / 100:7.1f} med {float(t.median()) / 100:7.1f} max {float(t.max()) / 100:7.1f}"
print(f"cin={CIN} res={RES} ablate={os.environ.get('SASPA_GEMM_ABLATE', '0')} event time {e0.elapsed_time(e1) * 1e3:.1f} us")
print("loop start (us after first) :", f(s0 - base))
print("loop end                    :", f(s1 - base))
print("kernel end                  :", f(end - base))
print("entry -> loop start (us)     :", f(pre))
print("loop duration               :", f(s1 - s0))
print("epilogue duration           :", f(end - s1))
entry, setup, issue, e0s, e1s, iss = st[:, 4], st[:, 5], st[:, 6], st[:, 7], st[:, 8], st[:, 9]
print("  entry -> im2col state     :", f(setup - entry))
print("  state -> prologue issued  :", f(issue - setup))
print("  issued -> loop start      :", f(s0 - issue))
print("  loop end -> epilogue start:", f(e0s - s1))
print("  half 0 (LDS + stores)     :", f(e1s - e0s))
print("  half 1 (LDS + stores)     :", f(iss - e1s))
print("  stores issued -> visible  :", f(end - iss))
print(f"in-loop clock GHz: med {float((clk.float() / (s1 - s0).float()).median()) * 0.1:.3f}")

"""Ablation + in-kernel clock of the software-pipelined flash attention loop (v3) at the level-0 shape; needs the
diagnostics library (`make -C saspa-aug_amd/csrc ABLATION=1`, SASPA_HIP_LIB=.../libsaspa_hip_abl.so).
bits: 1 no exponentials, 2 no MFMAs, 4 no LDS fragment reads, 8 no staging / barriers, 16 stamps."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
dev = torch.device('cuda:0')
B, H, N, D = 16, 8, 4096, 40
C = H * D
c = D ** -0.5 * 1.4426950408889634
qs = (torch.randn(B, N, C, device=dev) * c).bfloat16()
k = torch.randn(B, N, C, device=dev).bfloat16()
vt = torch.randn(B, C, N, device=dev).bfloat16()
nwg = (N // 128) * H * B           # enough for either workgroup size
full = torch.zeros(B * N * C + nwg * 8 + 64, device=dev, dtype=torch.bfloat16)
out = full[:B * N * C].view(B, N, C)
MODE = "4"
os.environ["SASPA_ATTN_MODE"] = MODE
print("SASPA_ATTN_MODE", MODE)
fl = 4.0 * B * H * N * N * D
for abl in (0, 16, 0, 1, 2, 4, 8, 14, 13, 11, 7):
    os.environ["SASPA_ATTN_ABLATE"] = str(abl)
    for _ in range(3):
        ops.flash_attn(qs, k, vt, out, H, D, N, N, 1.0, prescaled=True)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.flash_attn(qs, k, vt, out, H, D, N, N, 1.0, prescaled=True)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000 / 5)
    us = sorted(ts)[2]
    extra = ""
    if abl == 16:
        nw = nwg // 2
        st = full[B * N * C:B * N * C + nw * 8].view(torch.int64).view(nw, 2).cpu()
        clk = (st[:, 0].double() / st[:, 1].double() * 100.0)
        extra = f"  loop cycles median {int(st[:, 0].median())} ({int(st[:, 0].median()) / (N // 64):.0f} per 64-key step), in-kernel clock median {clk.median():.0f} MHz"
    print(f"abl={abl:2d}: {us:8.1f} us  ({fl / us / 1e6:6.1f} TF/s algorithmic){extra}", flush=True)

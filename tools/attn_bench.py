"""Flash-attention microbench over the shapes of the path (HIP events, 10 launches each).
SD-1.5 level 0/1/2 self- and cross-attention at CFG batch 16; SDXL levels 1/2 (head dim 64) at batch 8."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
dev = torch.device('cuda:0')
shapes = [(16, 8, 4096, 4096, 40), (16, 8, 1024, 1024, 80), (16, 8, 256, 256, 160), (16, 8, 4096, 77, 40), (16, 8, 1024, 77, 80),
          (8, 10, 1024, 1024, 64), (8, 20, 256, 256, 64), (8, 10, 4096, 4096, 64), (8, 10, 1024, 77, 64)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = [shapes[0], shapes[7], shapes[5]]
print("SASPA_ATTN_MODE =", os.environ.get("SASPA_ATTN_MODE"), flush=True)
for (B, H, NQ, NK, D) in shapes:
    C = H * D
    q = torch.randn(B, NQ, C, device=dev).bfloat16()
    k = torch.randn(B, NK, C, device=dev).bfloat16()
    vt = torch.randn(B, C, ops.round8(NK), device=dev).bfloat16()
    out = torch.empty(B, NQ, C, device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.flash_attn(q, k, vt, out, H, D, NQ, NK, D ** -0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.flash_attn(q, k, vt, out, H, D, NQ, NK, D ** -0.5)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"B={B} H={H} nq={NQ} nk={NK} d={D}: {us:8.1f} us  {4.0 * B * H * NQ * NK * D / us / 1e6:7.1f} TF/s", flush=True)

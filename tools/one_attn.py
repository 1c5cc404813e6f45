import sys, torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import saspa_aug_amd
from saspa_aug_amd import ops
dev = torch.device('cuda:0')
B, H, N, D = 16, 8, 4096, 40
C = H * D
qk = torch.randn(B, N, 2 * C, device=dev).bfloat16()
vt = torch.randn(B, C, N, device=dev).bfloat16()
out = torch.empty(B, N, C, device=dev, dtype=torch.bfloat16)
for _ in range(4): ops.flash_attn(qk[:, :, :C], qk[:, :, C:], vt, out, H, D, N, N, D ** -0.5)
torch.cuda.synchronize()

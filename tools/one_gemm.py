import sys, math, torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import saspa_aug_amd
from saspa_aug_amd import ops
dev = torch.device('cuda:0')
M, N, K = [int(a) for a in sys.argv[1:4]]
x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(5): ops.linear(x, w, out=out)
torch.cuda.synchronize()

import sys, math, torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import saspa_aug_amd
from saspa_aug_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for dtype in (torch.float32, torch.bfloat16):
    for (m, k, n) in [(16, 32, 16), (16, 64, 16), (64, 64, 64), (64, 128, 64), (128, 320, 128), (300, 320, 960)]:
        x = torch.randn(m, k).to(dtype).float(); w = (torch.randn(n, k) / math.sqrt(k)).to(dtype).float()
        ref = x @ w.t()
        out = ops.linear(x.to(dev, dtype), w.to(dev, dtype)).float().cpu()[:, :n]
        err = (out - ref).abs()
        # per-k-chunk probe: which k ranges contribute?
        print(dtype, (m, k, n), 'maxerr %.4g' % err.max().item(), 'ref max %.3g' % ref.abs().max().item())
    # one-hot probes: x = e_row, w = delta at k0 -> out[row][n0] should be 1
    k = 64 if dtype == torch.bfloat16 else 32
    for k0 in range(0, k, 4 if dtype == torch.float32 else 8):
        x = torch.zeros(16, k); w = torch.zeros(16, k)
        x[:, k0] = torch.arange(16) + 1.0; w[:, k0] = (torch.arange(16) + 1.0) * 100
        out = ops.linear(x.to(dev, dtype), w.to(dev, dtype)).float().cpu()
        ref = x @ w.t()
        ok = torch.equal(out, ref)
        if not ok:
            print(dtype, 'k0', k0, 'mismatch; out[1,:4]', out[1, :4].tolist(), 'ref[1,:4]', ref[1, :4].tolist(), 'out[:4,1]', out[:4, 1].tolist())
        else:
            print(dtype, 'k0', k0, 'ok')

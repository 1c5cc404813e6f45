import sys, math, torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import saspa_aug_amd
from saspa_aug_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 65536
for N in (160, 320, 1280):
    for K in (64, 128, 320, 640, 1280, 2560):
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        us = timeit(lambda: ops.linear(x, w, out=out))
        print(f"linear M={M} N={N} K={K}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s   bytes/us {(M*K*2+M*N*2)/us/1e6:.2f} TB/s")
# pure copy kernels for reference
a = torch.randn(M, 320, device=dev).bfloat16()
print("scale(copy) 42MB r + 42MB w:", timeit(lambda: ops.scale(a, 1.0)), "us")
print("torch copy:", timeit(lambda: a.clone()), "us")

"""HBM streaming rates as this box delivers them (context for the roofline of the store-bound epilogues): fill (write only),
sum (read only), copy (read + write) of buffers well beyond L2 + Infinity Cache, and of a 42 MB buffer (one conv output)."""
import torch
dev = torch.device('cuda:0')


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for mb in (42, 168, 1024, 4096):
    n = mb * (1 << 20) // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev)
    b = torch.empty(n, dtype=torch.bfloat16, device=dev)
    a.zero_(); b.zero_()
    tf = t(lambda: a.fill_(1.0))
    tc = t(lambda: b.copy_(a))
    ts = t(lambda: a.view(torch.int32).sum())
    by = n * 2
    print(f"{mb:5d} MiB: fill {by / tf / 1e12:5.2f} TB/s ({tf * 1e6:7.1f} us)   copy {2 * by / tc / 1e12:5.2f} TB/s r+w ({tc * 1e6:7.1f} us)   sum {by / ts / 1e12:5.2f} TB/s ({ts * 1e6:7.1f} us)")
    del a, b

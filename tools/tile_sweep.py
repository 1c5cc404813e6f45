import os, sys, math, subprocess
shapes = [(65536,320,320),(65536,320,1280),(65536,640,320),(16384,640,640),(16384,640,2560),(16384,1280,640),(4096,1280,1280),(4096,1280,5120),(4096,2560,1280),(1024,1280,1280),(1024,1280,5120),(1232,320,768),(1232,1280,768)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
    import saspa_aug_amd
    from saspa_aug_amd import ops
    dev = torch.device('cuda:0')
    def timeit(fn, n=40):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    res = []
    # cycle through distinct buffers so that operands are not cache-resident between calls
    for (M, N, K) in shapes:
        nbuf = max(2, int(600e6 // (M * (K + N) * 2)) )
        nbuf = min(nbuf, 24)
        xs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(nbuf)]
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
        i = [0]
        def f():
            j = i[0] % nbuf; i[0] += 1
            ops.linear(xs[j], w, out=outs[j])
        res.append(timeit(f))
    print(" ".join(f"{r:7.1f}" for r in res))
else:
    print("tile   " + " ".join(f"{m}x{n}x{k}"[:14].rjust(14) for m, n, k in shapes))
    for tile in ("0", "845"):
        env = dict(os.environ, SASPA_GEMM_TILE=tile)
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print(f"{tile:5s}  " + " ".join(v.rjust(14) for v in out.split()))

"""A/B of the K loops of the 8-wave wide GEMM (saspa_gemm_pp.hip) in ONE process, interleaved rounds: the two-barrier
ping-pong loop with four 20-MFMA phases per K-tile (SASPA_GEMM_PP_LOOP=0) against the one-barrier-per-phase asymmetric loop
(=1, round 3) and the two-barrier loop with two 40-MFMA phases per K-tile (=2, round 5), on the conv / linear shapes the
pipeline sends to that kernel; outputs must be bit-identical (same MFMA order per accumulator).
usage: python tools/pp_ab.py [704] [arms=pingpong,long]     (the `asym` arm needs the diagnostics library: `make ABLATION=1`,
SASPA_HIP_LIB=saspa-aug_amd/libsaspa_hip_abl.so -- the shipped library reads SASPA_GEMM_PP_LOOP=1 as the default loop)"""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops, weights as W
dev = torch.device('cuda:0'); BF = torch.bfloat16
# (batch, h, w, cin, cout, kind, ksplit): kind 3 = 3x3 conv, 1 = pointwise, 'u' = nearest x2 + 3x3
shapes = [(16, 64, 64, 320, 320, 3, None), (16, 64, 64, 640, 320, 3, None), (16, 64, 64, 960, 320, 3, None),
          (16, 32, 32, 640, 640, 3, None), (16, 32, 32, 1280, 640, 3, None), (16, 16, 16, 1280, 1280, 3, None),
          (16, 16, 16, 2560, 1280, 3, None), (16, 32, 32, 640, 640, 'u', None), (16, 16, 16, 1280, 1280, 'u', None),
          (16, 64, 64, 1280, 320, 1, None), (16, 16, 16, 1280, 10240, 'g', None),
          (8, 256, 256, 256, 256, 3, None), (8, 512, 512, 128, 128, 3, None), (8, 128, 128, 512, 512, 3, None)]
if "704" in sys.argv:
    shapes = [(16, 64, 88, 320, 320, 3, None), (16, 64, 88, 640, 320, 3, None), (16, 32, 44, 640, 640, 3, None),
              (16, 32, 44, 1280, 640, 3, None), (16, 16, 22, 1280, 1280, 3, None), (16, 16, 22, 2560, 1280, 3, None)]
ROUNDS, REP = 5, 10
ALL = {"pingpong": "0", "asym": "1", "long": "2"}
arms = [a for a in sys.argv[1:] if a.startswith("arms=")]
ARMS = [(n, ALL[n]) for n in (arms[0][5:].split(",") if arms else ["pingpong", "long"])]
for (b, h, w_, cin, cout, kind, ks) in shapes:
    taps = 1 if kind in (1, 'g') else 9
    K = taps * cin
    ho, wo = (2 * h, 2 * w_) if kind == 'u' else (h, w_)
    M = b * ho * wo
    xs = [torch.randn(b, h, w_, cin, device=dev).to(BF) for _ in range(3)]
    wt32 = torch.randn(cout, K) / math.sqrt(K)
    bias = torch.randn(cout, device=dev)
    if taps == 9:
        wt = W.to_chunk_major(wt32, 9, BF).to(dev, BF); wt.saspa_korder = 1
    elif kind == 'g':
        packed = W.pack_geglu(wt32, torch.randn(cout))
        wt, bias = packed[0].to(dev, BF), packed[1].to(dev, torch.float32)
    else:
        wt = wt32.to(dev, BF)
    i = [0]
    def f():
        j = i[0] % 3; i[0] += 1
        if taps == 9:
            return ops.conv(xs[j], wt, bias, kh=3, kw=3, pad=1, upsample=(kind == 'u'), variant=ops.GEMM_WIDE, ksplit=ks)
        if kind == 'g':
            return ops.linear(xs[j].view(-1, cin), wt, bias, act=ops.ACT_GEGLU, variant=ops.GEMM_WIDE)
        return ops.linear(xs[j].view(-1, cin), wt, bias, variant=ops.GEMM_WIDE)
    res, outs = {n: [] for n, _ in ARMS}, {}
    for rnd in range(ROUNDS + 1):
        for name, env in ARMS:
            os.environ["SASPA_GEMM_PP_LOOP"] = env
            i[0] = 0
            if rnd == 0:
                outs[name] = f().clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REP): f()
            e1.record(); torch.cuda.synchronize()
            if rnd: res[name].append(e0.elapsed_time(e1) * 1000 / REP)
    same = all(torch.equal(outs[ARMS[0][0]], outs[n]) for n, _ in ARMS[1:])
    fl = 2.0 * M * cout * K
    med = {n: sorted(v)[len(v) // 2] for n, v in res.items()}
    print(f"M={M} N={cout} K={K} kind={kind}: " + "  ".join(f"{n} {med[n]:7.1f} us {fl / med[n] / 1e6:6.0f} TF/s" for n in med)
          + f"  ratio {med[ARMS[0][0]] / med[ARMS[-1][0]]:.3f}  bit-identical={same}", flush=True)

"""`saspa_ff_block` against the two launches it replaces (A-stationary LayerNorm + GEGLU projection, output projection + residual)
at the level-0 shapes: M = 65 536 (512x512), 90 112 (512x704), 98 304 (512x768) tokens, C = 320, F = 1 280; HIP events, interleaved."""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops, weights as W
dev = torch.device('cuda:0'); BF = torch.bfloat16
f = 1280
w1 = torch.randn(2 * f, 320) / math.sqrt(320); b1 = torch.randn(2 * f) * 0.1
w2 = torch.randn(320, f) / math.sqrt(f); b2 = torch.randn(320) * 0.1
g, be = (1 + 0.1 * torch.randn(320)).to(dev), (0.1 * torch.randn(320)).to(dev)
w1p, b1p, w2f, b2p = [t.to(dev) for t in W.pack_ff_block(w1, b1, w2, b2)]
w1p, w2f = w1p.to(BF), w2f.to(BF)
wg, bg = W.pack_geglu(w1, b1); wg, bg = wg.to(dev, BF), bg.to(dev)
w2d, b2d = w2.to(BF).to(dev), b2.to(dev)
ABL = [int(a) for a in sys.argv[1:] if a.isdigit()]          # e.g. `python tools/ff_bench.py 1 2 4 8 16 31`: time the fused kernel under each ablation too
MS = [int(a[2:]) for a in sys.argv[1:] if a.startswith('m=')]           # e.g. m=32768 m=16384: other row counts
for m in (MS or (65536, 90112, 98304)) if not ABL else (65536,):
    xs = [torch.randn(m, 320, device=dev).to(BF) for _ in range(3)]
    outs = [torch.empty(m, 320, device=dev, dtype=BF) for _ in range(3)]
    i = [0]
    def fused_ws():
        j = i[0] % 3; i[0] += 1
        ops.ff_block(xs[j], (g, be, 1e-5), w1p, b1p, w2f, b2p, out=outs[j])
    def fused():                                             # the four-wave form (SASPA_FF_WS=0: read per launch)
        os.environ["SASPA_FF_WS"] = "0"
        fused_ws()
        os.environ.pop("SASPA_FF_WS", None)
    def pair():
        j = i[0] % 3; i[0] += 1
        hid = ops.linear(xs[j], wg, bg, act=ops.ACT_GEGLU, ln=(g, be, 1e-5)) if ops.linear_ln_fusable(xs[j], wg, act=ops.ACT_GEGLU) else \
            ops.linear(ops.layernorm(xs[j], g, be), wg, bg, act=ops.ACT_GEGLU)
        ops.linear(hid, w2d, b2d, residual=xs[j], out=outs[j])
    res = {"fused": [], "pair": [], "ws": []}
    for rnd in range(6):
        for name, fn in (("pair", pair), ("fused", fused), ("ws", fused_ws)):
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            if rnd: res[name].append(e0.elapsed_time(e1) * 100)
    med = lambda v: sorted(v)[len(v) // 2]
    fl = 2.0 * m * 320 * 2 * f + 2.0 * m * f * 320
    if "4wave" in sys.argv:                                  # ablations of the four-wave form (default: the wave-specialised one)
        os.environ["SASPA_FF_WS"] = "0"
    for a in ABL:
        os.environ["SASPA_FF_ABLATE"] = str(a)
        for _ in range(5): fused_ws()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fused_ws()
        e1.record(); torch.cuda.synchronize()
        print(f"  ablation {a:2d}: {e0.elapsed_time(e1) * 50:7.1f} us", flush=True)
    os.environ.pop("SASPA_FF_ABLATE", None)
    os.environ.pop("SASPA_FF_WS", None)
    print(f"M={m}: two launches {med(res['pair']):7.1f} us ({fl / med(res['pair']) / 1e6:5.0f} TF/s) | saspa_ff_block {med(res['ws']):7.1f} us ({fl / med(res['ws']) / 1e6:5.0f} TF/s) | its four-wave form {med(res['fused']):7.1f} us ({fl / med(res['fused']) / 1e6:5.0f} TF/s)", flush=True)

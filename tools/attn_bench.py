"""Flash-attention microbench over the shapes of the path (HIP events, interleaved rounds in ONE process).
SD-1.5 level 0/1/2 self- and cross-attention at CFG batch 16; SDXL levels 1/2 (head dim 64) at batch 8.
Columns: v1 = plain queries (scale applied per score); pre0 / pre1 / pre2 = prescaled queries (SASPA_ATTN_QPRESCALED)
through the v1 loop / the v2 loop with one LDS buffer / the v2 loop with two buffers and one barrier per tile
(SASPA_ATTN_MODE = 0 / 1 / 2); pre4 = the software-pipelined v3 loop, 8 waves per workgroup (mode 4); auto = the shipped dispatch rule;
rm = auto with V ROW-MAJOR (SASPA_ATTN_V_ROWMAJOR: the V columns of a fused Q | K | V buffer, transposed LDS reads); ring4 / rm4 = auto / rm
with SASPA_ATTN_RING=4 (v3 on its four-slot ring, one barrier per 64-key step, instead of six slots and one barrier per two steps); v3 = auto
with SASPA_ATTN_V4=0 (d = 40: the v3 loop instead of round 6's 64-queries-per-wave kernel; pre4 / ring4 are v3 too).  usage: python tools/attn_bench.py [quick] [512x704]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import ops
dev = torch.device('cuda:0')
shapes = [(16, 8, 4096, 4096, 40), (16, 8, 1024, 1024, 80), (16, 8, 256, 256, 160), (16, 8, 4096, 77, 40), (16, 8, 1024, 77, 80),
          (8, 10, 1024, 1024, 64), (8, 20, 256, 256, 64), (8, 10, 4096, 4096, 64), (8, 10, 1024, 77, 64)]
if "quick" in sys.argv:
    shapes = [shapes[0], shapes[7], shapes[5], shapes[1]]
if "512x704" in sys.argv:
    shapes = [(16, 8, 5632, 5632, 40), (16, 8, 1408, 1408, 80), (16, 8, 352, 352, 160), (16, 8, 5632, 77, 40)]
ROUNDS, REP = 5, 5
for (B, H, NQ, NK, D) in shapes:
    C = H * D
    c = D ** -0.5 * 1.4426950408889634
    q = torch.randn(B, NQ, C, device=dev).bfloat16()
    qs = (q.float() * c).bfloat16()
    k = torch.randn(B, NK, C, device=dev).bfloat16()
    vt = torch.randn(B, C, ops.round8(NK), device=dev).bfloat16()
    out = torch.empty(B, NQ, C, device=dev, dtype=torch.bfloat16)
    qkv = torch.randn(B, NK, 3 * C, device=dev).bfloat16()
    vrm = qkv[:, :, 2 * C:]
    vt.copy_(vrm.transpose(1, 2)[:, :, :NK]) if vt.shape[2] == NK else vt[:, :, :NK].copy_(vrm.transpose(1, 2))
    arms = [("v1", q, False, None), ("pre0", qs, True, "0"), ("pre2", qs, True, "2"), ("pre4", qs, True, "4"), ("auto", qs, True, ""),
            ("v3", qs, True, "v3"), ("ring4", qs, True, "ring4"), ("rm", qs, True, "rm"), ("rm4", qs, True, "rm4")]
    best = {a[0]: [] for a in arms}
    ref = None
    for rnd in range(ROUNDS + 1):
        for name, qq, pre, mode in arms:
            rm = mode in ("rm", "rm4")
            if mode in ("ring4", "rm4"):                      # the four-slot ring / one barrier per step (round 5's loop)
                os.environ["SASPA_ATTN_RING"] = "4"
            else:
                os.environ.pop("SASPA_ATTN_RING", None)
            if mode in ("v3", "ring4", "pre4"):               # without the 64-queries-per-wave kernel of round 6 (d = 40 only)
                os.environ["SASPA_ATTN_V4"] = "0"
            else:
                os.environ.pop("SASPA_ATTN_V4", None)
            if mode and mode.isdigit():
                os.environ["SASPA_ATTN_MODE"] = mode
            else:
                os.environ.pop("SASPA_ATTN_MODE", None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REP):
                ops.flash_attn(qq, k, vrm if rm else vt, out, H, D, NQ, NK, D ** -0.5, prescaled=pre, v_rowmajor=rm)
            e1.record(); torch.cuda.synchronize()
            if rnd:
                best[name].append(e0.elapsed_time(e1) * 1000 / REP)
            elif name == "v1":
                ref = out.float().clone()
            else:
                err = (out.float() - ref).abs().max().item()
                print(f"   {name}: max |out - v1| = {err:.3e}", flush=True)
    os.environ.pop("SASPA_ATTN_MODE", None)
    os.environ.pop("SASPA_ATTN_RING", None)
    os.environ.pop("SASPA_ATTN_V4", None)
    fl = 4.0 * B * H * NQ * NK * D
    line = "  ".join(f"{n} {sorted(v)[len(v) // 2]:8.1f} us {fl / sorted(v)[len(v) // 2] / 1e6:6.1f} TF/s" for n, v in best.items())
    print(f"B={B} H={H} nq={NQ} nk={NK} d={D}: {line}", flush=True)

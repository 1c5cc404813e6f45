"""Launch the dominant GEMM shapes of a UNet + ControlNet evaluation one by one (each twice: the SECOND dispatch of every
shape is the measured one) for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; tools/pmc_shapes_table.py turns the two
counter files into a per-shape table next to the algorithmic bytes.  SHAPES is shared by both scripts."""
import math, os, sys
SHAPES = [  # (kind, batch, H, W, Cin, Cout)  conv3x3 / linear (M = batch*H*W rows)
    ("conv", 16, 64, 64, 320, 320), ("conv", 16, 32, 32, 640, 640), ("conv", 16, 16, 16, 1280, 1280), ("conv", 16, 8, 8, 1280, 1280),
    ("linear", 16, 64, 64, 320, 320), ("linear", 16, 32, 32, 640, 640), ("linear", 16, 16, 16, 1280, 1280),
    ("geglu", 16, 64, 64, 320, 2560), ("geglu", 16, 32, 32, 640, 5120), ("geglu", 16, 16, 16, 1280, 10240),
    ("linear", 16, 64, 64, 1280, 320), ("linear", 16, 32, 32, 2560, 640), ("linear", 16, 16, 16, 5120, 1280),
]
if __name__ == "__main__":
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    import saspa_aug_amd  # noqa: F401
    from saspa_aug_amd import ops, weights as W
    dev = torch.device('cuda:0'); BF = torch.bfloat16
    for (kind, b, h, w_, cin, cout) in SHAPES:
        if kind == "conv":
            K = 9 * cin
            x = torch.randn(b, h, w_, cin, device=dev).to(BF)
            wt = W.to_chunk_major(torch.randn(cout, K) / math.sqrt(K), 9, BF).to(dev, BF); wt.saspa_korder = 1
            bias = torch.randn(cout, device=dev)
            f = lambda: ops.conv(x, wt, bias, kh=3, kw=3, pad=1)
        else:
            x = torch.randn(b * h * w_, cin, device=dev).to(BF)
            wf = torch.randn(cout, cin) / math.sqrt(cin); bf = torch.randn(cout)
            if kind == "geglu":
                wf, bf = W.pack_geglu(wf, bf)
            wt, bias = wf.to(dev, BF), bf.to(dev)
            res = None if kind == "geglu" else torch.randn(b * h * w_, cout, device=dev).to(BF)
            act = ops.ACT_GEGLU if kind == "geglu" else ops.ACT_NONE
            f = lambda: ops.linear(x, wt, bias, residual=res, act=act)
        f(); torch.cuda.synchronize(); f(); torch.cuda.synchronize()

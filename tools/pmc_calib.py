"""Which FETCH_SIZE correction applies to which GEMM kernel (roofline.traffic_by_kernel).

MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (128-B requests
tallied at 64 B) and is uncalibrated for other access shapes.  The LDS-DMA loaders of the GEMM kernels fetch 8 rows x 128 B per
wave instruction -- not the calibrated shape -- so this script launches, per kernel family, ONE problem whose beyond-L2 read
bytes are known by construction: a pointwise layer with M = 1 048 576 rows, K = 320 (A = 671 MB: larger than the 256 MiB
Infinity Cache plus the 32 MiB of L2, every row read exactly once by exactly one workgroup column), N = one tile's width
(160 / 320 columns: the weights are 0.1 - 0.2 MB), no residual.  FETCH_SIZE of that dispatch / (A + W bytes) = the factor the
counter applies to this kernel's reads: ~0.5 -> the x2 correction applies, ~1.0 -> the counter is exact.
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib_f -- python3 tools/pmc_calib.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/calib_w -- python3 tools/pmc_calib.py
    python3 tools/pmc_calib.py table <fetch csv> <write csv> [<out calibration.json>]  > profiles/rN_pmc_calibration.txt"""
import csv
import math
import os
import sys

CASES = [  # (label, variant, N, conv3x3)
    ("gemm_dma_kernel (4-wave 128x160, LDS-DMA)", 1, 160, False),
    ("gemm_pp_kernel (8-wave 256x320)", 2, 320, False),
    ("gemm_ws_kernel (12-wave wave-specialised)", 3, 160, False),
    ("gemm_as_kernel (A-stationary, K = 320)", 4, 320, False),
    ("gemm_pp_kernel 3x3 conv (chunk-major K)", 2, 320, True),
]
M, K = 1 << 20, 320

if len(sys.argv) > 1 and sys.argv[1] == "table":
    def rows(path, counter):
        out = []
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and "gemm" in r["Kernel_Name"]:
                out.append((int(r.get("Dispatch_Id", len(out))), r["Kernel_Name"], float(r["Counter_Value"]) * 1024))
        out.sort()
        return out
    fe, wr = rows(sys.argv[2], "FETCH_SIZE"), rows(sys.argv[3], "WRITE_SIZE")
    factors = {}
    print(f"{'kernel family':46s} {'A + W MB':>9s} {'FETCH raw MB':>13s} {'factor':>7s} {'out MB':>7s} {'WRITE MB':>9s} {'factor':>7s}  correction")
    # per case: the LAST dispatch of its kernel (each case launches twice, the second after the caches were flushed)
    def pick(rs, label, conv):
        fam = label.split(" ")[0]
        def is_pw(nm):
            return "<5, true" in nm or "Li5ELb1E" in nm or "ILi5ELb1" in nm
        cand = [r for r in rs if fam in r[1] and (fam != "gemm_pp_kernel" or is_pw(r[1]) != conv)]
        return cand[-1] if cand else None
    for i, (label, variant, n, conv) in enumerate(CASES):
        rf, rw = pick(fe, label, conv), pick(wr, label, conv)
        if rf is None or rw is None:
            print(f"{label:46s} (did not run)")
            continue
        mm = (1 << 18) if conv else M                                  # the conv case: 16 images of 128x128
        kk = 9 * K if conv else K
        inb = (mm * K + n * kk) * 2
        outb = mm * n * 2
        f, w = rf[2], rw[2]
        ff, wf = f / inb, w / outb
        if conv:
            corr = "x2 + halo rows re-fetched beyond L2 (not a clean calibration: its true bytes are not known by construction)"
        else:
            corr = "x2 (reads tallied at half)" if ff < 0.75 else "none (counter exact)"
        print(f"{label:46s} {inb / 1e6:9.1f} {f / 1e6:13.1f} {ff:7.3f} {outb / 1e6:7.1f} {w / 1e6:9.1f} {wf:7.3f}  {corr}   [{rf[1][:40]}]")
        factors[label.split(" ")[0] + (" conv" if conv else "")] = round(ff, 3)
    if len(sys.argv) > 4:
        import json
        # {kernel-name substring: measured FETCH_SIZE / known bytes}; the 3x3 case of the 8-wave kernel is listed for the record
        # {kernel-name substring: measured FETCH_SIZE / known bytes}.  The 3x3 instantiations of the 8-wave kernel get their own
        # entries (matched before the family's: tools/pmc_traffic_json.py takes the first key a name contains, most specific first)
        out = {k.replace(" conv", ""): v for k, v in factors.items() if " conv" not in k}
        if "gemm_pp_kernel conv" in factors:
            cv = factors["gemm_pp_kernel conv"]
            out = {"gemm_pp_kernel<5, false": cv, "gemm_pp_kernel<4, false": cv, "gemm_pp_kernelILi5ELb0": cv, "gemm_pp_kernelILi4ELb0": cv,
                   "conv_halo_kernel": cv, **out}
        json.dump(out, open(sys.argv[4], "w"), indent=1)
    sys.exit(0)

import torch  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401,E402
from saspa_aug_amd import ops, weights as W  # noqa: E402
dev = torch.device('cuda:0')
BF = torch.bfloat16
for (label, variant, n, conv) in CASES:
    if conv:
        x = torch.randn(16, 128, 128, K, device=dev).to(BF)
        wt = W.to_chunk_major(torch.randn(n, 9 * K) / math.sqrt(9 * K), 9, BF).to(dev, BF)
        wt.saspa_korder = 1
        f = lambda: ops.conv(x, wt, None, kh=3, kw=3, pad=1, variant=variant, ksplit=1)
    else:
        x = torch.randn(M, K, device=dev).to(BF)
        wt = (torch.randn(n, K) / math.sqrt(K)).to(dev, BF)
        f = lambda: ops.linear(x, wt, None, variant=variant, ksplit=1)
    try:
        f()
    except RuntimeError as e:            # a family that cannot take this problem: the table says so
        print(f"{label}: {e}", file=sys.stderr)
        del x, wt
        continue
    torch.cuda.synchronize()
    # evict: stream 1 GiB through the caches so that the measured dispatch reads A from HBM
    junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    junk.fill_(1)
    torch.cuda.synchronize()
    del junk
    f()
    torch.cuda.synchronize()
    del x, wt

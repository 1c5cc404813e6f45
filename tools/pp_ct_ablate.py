"""Compile-time ablation of the 8-wave GEMM's K loop: one library per ablation (no run-time switch inside the unrolled loop --
those change what they measure), timed on three conv shapes for each loop flavour.
  build (here, CPU):  python tools/pp_ct_ablate.py build      -> saspa-aug_amd/libsaspa_hip_ct<N>.so for N in the table below
  run (GPU box):      python tools/pp_ct_ablate.py
bits: 1 no MFMA · 2 no LDS fragment reads (MFMAs on stale registers) · 4 no DMA issue · 8 no barriers.  Results of an ablated
library are garbage; only the timing means anything."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "saspa-aug_amd", "csrc")
TABLE = (("full", 0), ("no LDS reads", 2), ("no DMA", 4), ("no reads, no DMA (MFMA + barriers)", 6), ("no barriers", 8),
         ("MFMA only", 14), ("no MFMA", 1), ("barriers only", 7), ("stamps: wait behind an MFMA block (loop 2)", 32),
         ("stamps, no DMA", 36), ("stamps, no reads", 34), ("no s_setprio around the MFMA block (loop 2)", 64),
         ("stamps + no s_setprio", 96))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    objs = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".o") and ".abl." not in f and not f.startswith("saspa_gemm_pp")]
    only = [int(a) for a in sys.argv[2:]]
    for _, n in TABLE:
        if n == 0 or (only and n not in only):
            continue
        o = os.path.join(CSRC, f"saspa_gemm_pp.ct{n}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT}/include",
                               f"-DSASPA_PP_CT_ABL={n}", "-c", os.path.join(CSRC, "saspa_gemm_pp.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", o] + [os.path.join(CSRC, f) for f in objs]
                              + ["-o", os.path.join(ROOT, "saspa-aug_amd", f"libsaspa_hip_ct{n}.so")])
        print("built", n, flush=True)
elif len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    import saspa_aug_amd  # noqa: F401
    from saspa_aug_amd import ops, weights as W
    dev = torch.device('cuda:0')

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n * 1e3)
        return best
    out, extra = [], []
    for (b, h, w_, cin, cout) in ((16, 64, 64, 320, 320), (16, 64, 64, 960, 320), (16, 32, 32, 1280, 640)):
        x = torch.randn(b, h, w_, cin, device=dev).bfloat16()
        wt = W.to_chunk_major(torch.randn(cout, 9 * cin) / (3 * cin ** 0.5), 9, torch.bfloat16).to(dev, torch.bfloat16)
        wt.saspa_korder = 1
        o = torch.empty(b, h, w_, cout, device=dev, dtype=torch.bfloat16)
        out.append(timeit(lambda: ops.conv(x, wt, None, kh=3, kw=3, pad=1, variant=ops.GEMM_WIDE, out=o)))
        import re
        m_ = re.search(r"_ct(\d+)\.so", os.environ.get("SASPA_HIP_LIB", ""))
        if m_ and (int(m_.group(1)) & 32) and os.environ.get("SASPA_GEMM_PP_LOOP") == "2":
            torch.cuda.synchronize()
            d = o.view(-1)[:16].view(torch.int32).cpu().tolist()      # waves 0 / 4 of workgroup 0: {sum Q0, sum Q1, loop, intervals}
            for g in (0, 1):
                q0, q1, loop, n = d[4 * g:4 * g + 4]
                n = max(n, 1)
                extra.append(f"[group {g}: wait behind a Q0 block {2 * q0 / n:.0f}, Q1 block {2 * q1 / n:.0f}, loop {loop / n:.0f} cycles per interval]")
    print(" ".join(f"{v:9.1f}" for v in out) + " " + " ".join(extra))
else:
    print("variant                                  loop   conv 65536x320x2880   65536x320x8640   16384x640x11520  (us)")
    only = [int(a) for a in sys.argv[1:]]
    for name, n in TABLE:
        if only and n not in only:
            continue
        lib = os.path.join(ROOT, "saspa-aug_amd", f"libsaspa_hip_ct{n}.so" if n else "libsaspa_hip.so")
        for loop in ("2", "0"):
            env = dict(os.environ, SASPA_HIP_LIB=lib, SASPA_GEMM_PP_LOOP=loop)
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            line = (r.stdout.strip().splitlines() or ["(failed) " + r.stderr.strip()[-200:]])[-1]
            print(f"{name:40s} {loop:>4s}   {line}", flush=True)

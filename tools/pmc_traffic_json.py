"""HBM traffic per launch of the implicit-GEMM kernel family from rocprofv3 --pmc passes over tools/pmc_step.py.
usage: pmc_traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv> [<calibration.json>] > profiles/..._pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 reports 1/2 of the bytes of wide (16 B/lane) streaming reads (MI355X_MICROARCH.md, HBM /
rocprofv3 section); whether that applies to a kernel's reads is measured per kernel family by tools/pmc_calib.py (a problem whose
beyond-L2 read bytes are known by construction): `calibration.json` = {family substring: fetch factor}.  Output: the family
totals with the read side uncorrected (`raw`, a lower bound), doubled everywhere (`x2`, an upper bound) and corrected per kernel
by its calibrated factor (`best`), plus the same three per kernel name (`by_kernel`)."""
import csv, json, re, sys
FAMILY = ("gemm_dma_kernel", "gemm_pp_kernel", "gemm_kernel", "gemm_ws_kernel", "gemm_as_kernel", "xattn_block_kernel", "conv_halo_kernel")
calib = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}
def short(name):
    return re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+|void ", "", name)[:64]
def per_kernel(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and any(f in r["Kernel_Name"] for f in FAMILY):
            d = out.setdefault(short(r["Kernel_Name"]), [0.0, 0])
            d[0] += float(r["Counter_Value"]) * 1024; d[1] += 1
    return out
fe, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
def factor(name):
    for k, v in calib.items():
        if k in name:
            return float(v)
    return None
by_kernel, tot = {}, dict(n=0, raw=0.0, x2=0.0, best=0.0, w=0.0)
for name, (fb, n) in sorted(fe.items(), key=lambda kv: -kv[1][0]):
    wb = wr.get(name, [0.0, 0])[0]
    f = factor(name)
    # `best`: the read side divided by the factor MEASURED for the kernel (0.50 - 0.54 for the LDS-DMA readers of pointwise
    # operands = the guide's x2; 0.77 for the 3x3 instantiations of the 8-wave kernel, whose reads are 128-byte rows the counter
    # tallies closer to their size); uncalibrated kernels: the guide's x2
    corr = (1.0 / f if f > 0.05 else 2.0) if f is not None else 2.0
    by_kernel[name] = dict(launches=n, fetch_raw_MB=round(fb / n / 1e6, 2), write_MB=round(wb / n / 1e6, 2),
                           fetch_factor_measured=f, correction=(f"x{corr:.2f}") + ("" if f is not None else " (uncalibrated)"),
                           hbm_MB_per_launch=round((corr * fb + wb) / n / 1e6, 2))
    tot["n"] += n; tot["raw"] += fb; tot["x2"] += 2 * fb; tot["best"] += corr * fb; tot["w"] += wb
n = max(tot["n"], 1)
out = {"kernel_family": " / ".join(FAMILY) + " (all instantiations)", "workload": "tools/pmc_step.py: warm + 2-step batch-8 512x512 generation",
       "launches": tot["n"], "fetch_bytes_per_launch_raw": tot["raw"] / n, "fetch_bytes_per_launch_x2": tot["x2"] / n,
       "write_bytes_per_launch": tot["w"] / n, "hbm_bytes_per_launch": (tot["x2"] + tot["w"]) / n,
       "hbm_bytes_per_launch_best": (tot["best"] + tot["w"]) / n, "calibration": calib, "by_kernel": by_kernel}
print(json.dumps(out, indent=1))

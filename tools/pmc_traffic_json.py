"""HBM traffic per launch of the implicit-GEMM kernel family from rocprofv3 --pmc passes over tools/pmc_step.py.
usage: pmc_traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/..._pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are KiB; gfx950 reports 1/2 of the bytes of wide (16 B/lane) streaming reads, so the read side
is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv, json, sys
def total(path, counter):
    s, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and ("gemm_dma_kernel" in r["Kernel_Name"] or "gemm_pp_kernel" in r["Kernel_Name"]
                                             or "gemm_kernel" in r["Kernel_Name"] or "gemm_ws_kernel" in r["Kernel_Name"]
                                             or "gemm_as_kernel" in r["Kernel_Name"]):
            s += float(r["Counter_Value"]); n += 1
    return s, n
f, nf = total(sys.argv[1], "FETCH_SIZE")
w, nw = total(sys.argv[2], "WRITE_SIZE")
out = {"kernel_family": "gemm_dma_kernel / gemm_pp_kernel / gemm_ws_kernel / gemm_as_kernel / gemm_kernel (all instantiations)", "workload": "tools/pmc_step.py: warm + 2-step batch-8 512x512 generation",
       "launches": nf, "fetch_bytes_per_launch_raw": f * 1024 / max(nf, 1), "fetch_bytes_per_launch_x2": 2 * f * 1024 / max(nf, 1),
       "write_bytes_per_launch": w * 1024 / max(nw, 1), "hbm_bytes_per_launch": (2 * f + w) * 1024 / max(nf, 1)}
print(json.dumps(out, indent=1))

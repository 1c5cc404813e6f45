"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (summary only).
usage: python tools/gap_summary.py <dir with *_kernel_trace.csv> [skip_first_n]"""
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
# keep the densest region: drop leading part before the last long (>5 ms) gap
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 5_000_000: cut = i
rows = rows[cut:]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = [max(0, rows[i][0] - rows[i - 1][1]) for i in range(1, len(rows))]
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  sum of durations {busy / 1e6:.2f} ms  sum of gaps {sum(gaps) / 1e6:.2f} ms ({100 * sum(gaps) / span:.1f} % of span)")
gs = sorted(gaps)
print("gap ns: median %d  p90 %d  p99 %d  max %d" % (gs[len(gs) // 2], gs[int(len(gs) * .9)], gs[int(len(gs) * .99)], gs[-1]))
by = collections.defaultdict(lambda: [0, 0, 0])
for i in range(1, len(rows)):
    k = rows[i][2][:70]
    by[k][0] += 1; by[k][1] += gaps[i - 1]; by[k][2] += rows[i][1] - rows[i][0]
print("gap BEFORE kernel (by following kernel), top 12 by total gap:")
for k, (n, g, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {k:70s} n {n:6d} avg gap {g / n / 1e3:7.2f} us  avg dur {d / n / 1e3:8.2f} us")

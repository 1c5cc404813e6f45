"""How busy is the GPU during the hipGraph replay of the sampling loop?  From a rocprofv3 --kernel-trace run of bench.py: the
union of all kernel intervals (the two graph branches overlap) against the wall-clock span of the densest 1-second window, the
idle time (no kernel at all running) and its distribution, and what follows the idle gaps.
usage: python tools/busy_summary.py <dir with *_kernel_trace.csv> [window_ms]"""
import collections, csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
win = (float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0) * 1e6
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# densest window of `win` ns by kernel count (the timed graph replays: ~60 000 kernels per second)
best, j = (0, 0), 0
for i in range(len(rows)):
    while rows[i][0] - rows[j][0] > win: j += 1
    if i - j > best[0]: best = (i - j, j)
n, j = best
seg = rows[j:j + n]
span = seg[-1][1] - seg[0][0]
busy, idle_gaps, cur_end = 0, [], seg[0][0]
after = collections.defaultdict(lambda: [0, 0])
for s, e, k in seg:
    if s > cur_end:
        idle_gaps.append(s - cur_end)
        a = after[k[:60]]; a[0] += 1; a[1] += s - cur_end
        busy += 0
        cur_start = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
tot = sum(e - s for s, e, _ in seg)
print(f"window: {n} kernels, span {span / 1e6:.1f} ms; sum of kernel durations {tot / 1e6:.1f} ms ({100 * tot / span:.1f} % of span: branches overlap)")
print(f"GPU busy (union of kernel intervals) {busy / 1e6:.1f} ms = {100 * busy / span:.1f} % of span; idle {100 - 100 * busy / span:.1f} % in {len(idle_gaps)} gaps")
g = sorted(idle_gaps)
if g:
    print("idle gap ns: median %d  p90 %d  p99 %d  max %d; kernels per idle gap %.1f" % (g[len(g) // 2], g[int(len(g) * .9)], g[int(len(g) * .99)], g[-1], n / len(g)))
print("idle time by the kernel that ends it (top 10):")
for k, (c, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {k:60s} n {c:6d}  total {t / 1e6:7.2f} ms  avg {t / c / 1e3:6.2f} us")

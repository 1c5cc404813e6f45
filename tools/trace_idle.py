"""How much of a generation's wall time has NO kernel on the GPU?  Reads a rocprofv3 --kernel-trace CSV of a bench run, takes
the last `window` seconds of it (default 1.0: inside the last timed batch, all sampling steps), merges the kernels' [start, end]
intervals and reports: busy / idle share, the histogram of the idle gaps, the kernels that most often precede a gap, and the
share of time with two or more kernels in flight (the paired encoder region).
usage: python tools/trace_idle.py <kernel_trace.csv> [window_seconds]"""
import collections
import csv
import sys

path = sys.argv[1]
window = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
ev = []
for r in csv.DictReader(open(path)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
ev.sort()
t_end = max(e[1] for e in ev)
# the tail of the run is the VAE decode + safety checker of the last batch: step back to the last flash-attention launch of the
# level-0 shape (a sampling-step kernel) and take the window before it
last_fa = max(e[1] for e in ev if "flash_attn_v3" in e[2])
t1 = last_fa
t0 = t1 - int(window * 1e9)
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print(f"{len(win)} kernels in a {window:.2f} s window ending at the last sampling-step attention launch")
busy = 0
gaps = []
cur_s, cur_e = win[0][0], win[0][1]
last_name = win[0][2]
# `depth` time: sweep line over starts / ends
pts = []
for s, e, _ in win:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth_t = collections.Counter()
d, prev = 0, pts[0][0]
for t, k in pts:
    depth_t[min(d, 3)] += t - prev
    prev = t; d += k
for s, e, name in win[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, name))
        cur_s, cur_e = s, e
        last_name = name
    else:
        if e > cur_e:
            cur_e = e
            last_name = name
busy += cur_e - cur_s
span = win[-1][1] - win[0][0]
idle = span - busy
print(f"span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms ({100 * busy / span:.2f} %), idle {idle / 1e6:.1f} ms ({100 * idle / span:.2f} %) in {len(gaps)} gaps")
print("time with k kernels in flight: " + ", ".join(f"k={k}{'+' if k == 3 else ''}: {100 * v / span:.1f} %" for k, v in sorted(depth_t.items())))
edges = [1, 2, 3, 5, 10, 20, 50, 100, 1000, 1e9]
hist = collections.Counter()
hsum = collections.Counter()
for g, _, _ in gaps:
    for e in edges:
        if g / 1e3 <= e:
            hist[e] += 1; hsum[e] += g
            break
print("gap length (us)      count   total ms   share of span")
lo = 0
for e in edges:
    print(f"  {lo:>6g} - {e:<8g} {hist[e]:7d} {hsum[e] / 1e6:10.2f} {100 * hsum[e] / span:10.2f} %")
    lo = e


def short(n):
    import re
    return re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+|void ", "", n)[:60]


by_prev = collections.Counter()
for g, a, _ in gaps:
    by_prev[short(a)] += g
print("idle time by the kernel that ended before the gap (top 12):")
for n, v in by_prev.most_common(12):
    print(f"  {n:62s} {v / 1e6:8.2f} ms")
by_next = collections.Counter()
for g, _, b in gaps:
    by_next[short(b)] += g
print("idle time by the kernel that started after the gap (top 12):")
for n, v in by_next.most_common(12):
    print(f"  {n:62s} {v / 1e6:8.2f} ms")

import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
a = torch.randn(2048, 2048); b = torch.randn(2048, 2048)
x = torch.randn(2, 320, 64, 64); w = torch.randn(320, 320, 3, 3)
for nt in (4, 8, 16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1): break
    torch.set_num_threads(nt)
    a @ b
    t = time.time(); n = 0
    while time.time() - t < 0.5: a @ b; n += 1
    mm = n * 2 * 2048**3 / (time.time() - t) / 1e12
    torch.nn.functional.conv2d(x, w, padding=1)
    t = time.time(); n = 0
    while time.time() - t < 0.5: torch.nn.functional.conv2d(x, w, padding=1); n += 1
    cv = n * 2 * 2 * 4096 * 320 * 2880 / (time.time() - t) / 1e12
    print(f"threads {nt}: matmul {mm:.3f} TFLOP/s conv {cv:.3f} TFLOP/s")

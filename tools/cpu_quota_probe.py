import os, time, threading, zlib
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a")
for f in ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
buf = os.urandom(4 << 20)
def w(): zlib.crc32(buf); [zlib.adler32(buf) for _ in range(40)]
for n in (1, 8, 16, 32, 64, 128):
    ts = [threading.Thread(target=w) for _ in range(n)]; t = time.perf_counter(); [x.start() for x in ts]; [x.join() for x in ts]
    dt = time.perf_counter() - t; print(n, "threads: %.0f ms -> %.1f thread-equivalents" % (dt * 1e3, n * base / dt if n > 1 else 1.0)) if n > 1 else None
    if n == 1: base = dt; print("1 thread: %.0f ms" % (dt * 1e3))

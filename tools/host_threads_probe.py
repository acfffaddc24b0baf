"""What the container may use of its host, and what torch makes of it: affinity mask, cgroup CFS quota, torch pool size, and an
fp32 / bf16 Linear of the oracle's prefill shape (243 x 4096 x 11008) at 8 .. 256 threads.  Round 6: the GPU boxes have a 16-CPU quota on a
256-thread host and torch defaults to 128 threads there (tests/conftest.py sizes the pool from the quota).
    gpurun -- python tools/host_threads_probe.py"""
import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "n/a")
try: print("loadavg", open("/proc/loadavg").read().strip())
except Exception: pass
x32 = torch.randn(243, 4096); w32 = torch.randn(11008, 4096)
xb = x32.bfloat16(); wb = w32.bfloat16()
for n in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(n)
    for name, (a, b) in (("fp32", (x32, w32)), ("bf16", (xb, wb))):
        torch.nn.functional.linear(a, b)
        t = time.perf_counter()
        for _ in range(5): torch.nn.functional.linear(a, b)
        dt = (time.perf_counter() - t) / 5
        print(f"threads {n:3d} {name}: {dt*1e3:7.2f} ms  {2*243*4096*11008/dt/1e9:7.1f} GFLOP/s")

"""Do two HIP streams of this process run kernels concurrently on this box?  (env + a timing probe)"""
import os, time, torch
print({k: v for k, v in os.environ.items() if any(s in k for s in ("GPU_", "HIP_", "HSA_", "ROCR", "AMD_"))})
x = torch.randn(64, 1 << 14, device="cuda")          # small grid: 64 workgroups' worth of work per kernel
def work(t, n=200):
    for _ in range(n):
        t = torch.sin(t)
    return t
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for prio in (None, -1):
    if prio is not None:
        s2 = torch.cuda.Stream(priority=prio)
    work(x); torch.cuda.synchronize()
    t0 = time.perf_counter(); work(x); work(x); torch.cuda.synchronize(); serial = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.cuda.stream(s1): work(x)
    with torch.cuda.stream(s2): work(x)
    torch.cuda.synchronize(); par = time.perf_counter() - t0
    print(f"priority {prio}: serial {serial*1e3:.2f} ms, two streams {par*1e3:.2f} ms")

"""Streaming-read calibration on the GPU box: what a trivial kernel reaches on this MI355X (the denominator the
roofline fractions in DESIGN.md should be read against, next to the 8 TB/s datasheet peak)."""
import time
import torch

def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

for gb in (1.5, 3.0, 6.0):
    n = int(gb * 1e9 / 2)
    a = torch.ones(n, dtype=torch.float16, device="cuda")
    t = timeit(lambda: a.sum(dtype=torch.float32))
    print(f"read  {gb:.1f} GB fp16 sum      : {gb / t / 1e3:.2f} TB/s ({t*1e3:.3f} ms)")
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    print(f"copy  {gb:.1f} GB (read+write)  : {2 * gb / t / 1e3:.2f} TB/s ({t*1e3:.3f} ms)")
    a32 = a.view(torch.float32)
    t = timeit(lambda: a32.max())
    print(f"read  {gb:.1f} GB fp32 max      : {gb / t / 1e3:.2f} TB/s ({t*1e3:.3f} ms)")
    del a, b, a32

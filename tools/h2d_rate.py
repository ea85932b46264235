"""What PCIe gives a 256^3 u8 time step (16 MiB, pinned host memory -> device): one copy, two halves and four quarters on two streams.
Measured: 55.2 / 53.4 / 55.6 GB/s -- the link, not the copy engine, is the limit: cpm_volume_stream keeps ONE copy per step.
usage: python tools/h2d_rate.py"""
import torch, time
n = 16 << 20
src = torch.empty(n, dtype=torch.uint8).pin_memory()
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    with torch.cuda.stream(s1):
        dst.copy_(src, non_blocking=True)
def two(parts=2):
    streams = [s1, s2]
    step = n // parts
    for p in range(parts):
        with torch.cuda.stream(streams[p % 2]):
            dst[p * step:(p + 1) * step].copy_(src[p * step:(p + 1) * step], non_blocking=True)
for name, fn in (("one copy", one), ("two halves on two streams", two), ("four quarters on two streams", lambda: two(4))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 50
    print(f"{name}: {dt * 1e3:.3f} ms = {n / dt / 1e9:.1f} GB/s")

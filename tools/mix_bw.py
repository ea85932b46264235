"""Streaming-kernel check: achieved HBM bandwidth of the temporal mix kernels at sizes past the Infinity Cache."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
B = cpm_amd.binding
ctx = B.Context(0)
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
import ctypes
for cap in (-1, 8, 0):
  ctx.lib.cpm_debug_set_stream_wg_per_cu.argtypes=[ctypes.c_void_p, ctypes.c_int]; ctx.lib.cpm_debug_set_stream_wg_per_cu(ctx.h, cap); print('wg/CU', cap)
  for n in (1 << 28,):
    x = torch.rand(n, device='cuda'); y = torch.rand(n, device='cuda'); o = torch.empty_like(x)
    t = timeit(lambda: ctx.mix_buffers(x, y, 0.3, o))
    print(f"mix_f32 n={n}: {t*1e6:.1f} us  {3 * 4 * n / t / 1e12:.2f} TB/s")
    t = timeit(lambda: torch.lerp(x, y, 0.3, out=o))
    print(f"  torch.lerp       : {t*1e6:.1f} us  {3 * 4 * n / t / 1e12:.2f} TB/s")
    del x, y, o
  for dim in (256, 1024):
    a = np.zeros((dim, dim, dim), np.uint8)
    va, vb, vo = ctx.volume_create(a), ctx.volume_create(a), ctx.volume_create(a)
    t = timeit(lambda: ctx.volume_mix(va, vb, 0.3, vo))
    print(f"volume_mix u8 {dim}^3: {t*1e6:.1f} us  {3 * dim**3 / t / 1e12:.2f} TB/s")
    del va, vb, vo

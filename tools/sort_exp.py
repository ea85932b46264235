import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
B = cpm_amd.binding
ctx = B.Context(0)
ctx.lib.cpm_debug_set_sort_items.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = 1 << 20
rng = np.random.default_rng(0)
keys = torch.from_numpy(rng.integers(0, 1 << 22, n).astype(np.int32)).cuda()
vals = torch.arange(n, dtype=torch.int32, device='cuda')
for items in (0, 8, 16):
    ctx.lib.cpm_debug_set_sort_items(ctx.h, items)
    for bits in (22, 31):
        k, v = keys.clone(), vals.clone()
        for _ in range(3): ctx.sort_pairs(k, v, bits)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): ctx.sort_pairs(k, v, bits)
        torch.cuda.synchronize()
        print(f"items {items or 4} bits {bits}: {(time.perf_counter() - t) / 50 * 1e6:.1f} us")

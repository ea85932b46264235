"""Experiment: where does the gather's time go? (run on the GPU box)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), S.workspace_tf(), 1024, (128,)*3, light_travel_direction=(0.3, 0.5, -1.0))
fr.frame(); torch.cuda.synchronize()

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6

print('gather real        us', timeit(fr.gather))
cs_real = fr.cell_start.clone()
fr.cell_start.zero_()
print('gather empty grid  us', timeit(fr.gather))
fr.cell_start.copy_(cs_real)
# only the z-face photons: zero out everything except top 8 slabs? emulate by making photons uniform
n = fr.n
rng = np.random.default_rng(0)
ph = np.zeros((n, 8), np.float32); ph[:, :3] = rng.random((n, 3), dtype=np.float32); ph[:, 3:6] = 1
fr.photons.copy_(torch.from_numpy(ph).to(ctx.device))
fr.bin(); torch.cuda.synchronize()
print('gather uniform photons (0.5/cell) us', timeit(fr.gather))
print('bin uniform us', timeit(fr.bin))
print('splat uniform us', timeit(lambda: fr.splat()))

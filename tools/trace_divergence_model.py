"""Divergence model of the tracer from the oracle's per-photon Woodcock iteration counts (CPU only).

What a wave costs today is the iteration count of its slowest lane; this prints the distribution and what
re-packing unfinished photons after K iterations would leave."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import ctypes as C
import numpy as np, cpm_amd
from oracle_binding import Oracle, OTraceParams
S, P = cpm_amd.synthetic, cpm_amd.pipeline
o = Oracle()
vol_np, tf = S.heterogeneous_volume(256), S.workspace_tf()
nx = ny = 1024; n = nx * ny
d = P._normalize((0.3, 0.5, -1.0))
origin = np.array([0.5] * 3, np.float32) - np.float32(2.0) * d
po_, u, v = P.fit_plane_aligned_obb(S.UNIT_CUBE_VERTICES, origin, d)
area = float(np.float32(np.linalg.norm(u)) * np.float32(np.linalg.norm(v)))
s = o.uniform_samples_2d(nx, ny); ls = o.directional_light_samples(s, (1, 1, 1), d, po_, u, v, area)
isect = o.light_sample_box_intersection(ls, S.UNIT_CUBE_AABB)
st = np.zeros((n, 2), np.uint32); st[:, 0] = o.glibc_rand_sequence(0, n); o.seed_streams(st, 1 << 40)
ovol = o.volume(vol_np)
p = OTraceParams(); p.step_size = 1.0 / 256; p.n_light_samples = n; p.max_interactions = 1; p.total_photons = n
photons = np.zeros((n, 8), np.float32)
steps = np.zeros(n, np.uint32)
o.lib.cpmo_debug_set_step_array.argtypes = [C.c_void_p]
o.lib.cpmo_debug_set_step_array(steps.ctypes.data)
o.trace(ovol, tf, S.UNIT_CUBE_AABB, p, ls, isect, st, photons)
o.lib.cpmo_debug_set_step_array(None)
miss = isect[:, 0] >= isect[:, 1]
print('photons', n, 'missing the box', int(miss.sum()), 'mean steps', steps.mean(), 'max', steps.max())
print('histogram (steps: photons):', {k: int((steps == k).sum()) for k in range(0, 12)}, '>=12:', int((steps >= 12).sum()))
w = steps.reshape(-1, 64)
wm = w.max(axis=1)
print('waves', len(wm), 'mean of wave max', wm.mean(), 'sum of wave max', int(wm.sum()), '= lane utilisation', steps.sum() / (64.0 * wm.sum()))
print('wave max histogram:', {k: int((wm == k).sum()) for k in range(0, 40, 1) if (wm == k).any()})
for K in (1, 2, 3, 4, 6, 8):
    first = np.minimum(wm, K).sum()
    rest = np.sort(steps[steps > K] - K)  # repacked in any order: lower bound with sorted packing; random packing below
    left = steps[steps > K] - K
    pad = (-len(left)) % 64
    rnd = np.concatenate([left, np.zeros(pad, left.dtype)]).reshape(-1, 64).max(axis=1).sum()
    print(f'K={K}: pass 1 wave-iterations {int(first)}, survivors {len(left)} ({len(left)/n:.3f}), pass 2 wave-iterations (index order) {int(rnd)}, total {int(first + rnd)} vs {int(wm.sum())}')

"""CPU baseline tuning aid: oracle frame time against the OpenMP thread count."""
import sys, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, cpm_amd
from oracle_binding import Oracle, OTraceParams
S, P = cpm_amd.synthetic, cpm_amd.pipeline
o = Oracle()
vol_np, tf = S.heterogeneous_volume(256), S.workspace_tf()
nx = ny = 1024; n = nx * ny
d = P._normalize((0.3, 0.5, -1.0))
origin = np.array([0.5]*3, np.float32) - np.float32(2.0) * d
po_, u, v = P.fit_plane_aligned_obb(S.UNIT_CUBE_VERTICES, origin, d)
area = float(np.float32(np.linalg.norm(u)) * np.float32(np.linalg.norm(v)))
s = o.uniform_samples_2d(nx, ny); ls = o.directional_light_samples(s, (1, 1, 1), d, po_, u, v, area)
isect = o.light_sample_box_intersection(ls, S.UNIT_CUBE_AABB)
st = np.zeros((n, 2), np.uint32); st[:, 0] = o.glibc_rand_sequence(0, n); o.seed_streams(st, 1 << 40)
ovol = o.volume(vol_np); og = o.grid((128,)*3, 1)
p = OTraceParams(); p.step_size = 1.0/256; p.n_light_samples = n; p.max_interactions = 1; p.total_photons = n
photons = np.zeros((n, 8), np.float32); out = np.zeros(128**3, np.float32)
radius = S.photon_radius_texture((256,)*3, 1.0); scale = o.relative_irradiance_scale(radius, n)
for T in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "16,32,64,128,256".split(","))]:
    o.set_threads(T)
    best = None
    for rep in range(4):
        t0 = time.perf_counter(); o.trace(ovol, tf, S.UNIT_CUBE_AABB, p, ls, isect, st, photons)
        t1 = time.perf_counter(); _, cs, srt = o.bin(photons, n, og)
        t2 = time.perf_counter(); o.gather(srt, cs, n, og, radius, scale, out)
        t3 = time.perf_counter()
        if rep and (best is None or t3 - t0 < best[0]): best = (t3 - t0, t1 - t0, t2 - t1, t3 - t2)
    print(T, 'threads: frame %.3f s = trace %.3f + bin %.3f + gather %.3f' % best, flush=True)

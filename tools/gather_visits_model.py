"""Record-visit model of the brick gather on config 2's photons (CPU, oracle trace): how many
(record, brick) visits a voxel-brick shape costs, counting every record in a brick's halo once.
usage: python tools/gather_visits_model.py"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle_binding import Oracle, OTraceParams
import cpm_amd
P, S = cpm_amd.pipeline, cpm_amd.synthetic
o = Oracle(); o.set_threads(8)
nx = ny = 1024; n = nx * ny; G = 128
vol = S.heterogeneous_volume(256); tf = S.workspace_tf()
d = P._normalize((0.3, 0.5, -1.0))
origin = np.array([0.5, 0.5, 0.5], np.float32) - np.float32(2.0) * d
po_, u, v = P.fit_plane_aligned_obb(S.UNIT_CUBE_VERTICES, origin, d)
area = float(np.float32(np.linalg.norm(u)) * np.float32(np.linalg.norm(v)))
ls = o.directional_light_samples(o.uniform_samples_2d(nx, ny), (1, 1, 1), d, po_, u, v, area)
isect = o.light_sample_box_intersection(ls, S.UNIT_CUBE_AABB)
st = np.zeros((n, 2), np.uint32); st[:, 0] = o.glibc_rand_sequence(0, n); o.seed_streams(st, 1 << 40)
p = OTraceParams(); p.step_size = 1.0 / 256; p.n_light_samples = n; p.max_interactions = 1; p.total_photons = n
ph = np.zeros((n, 8), np.float32)
o.trace(o.volume(vol), tf, S.UNIT_CUBE_AABB, p, ls, isect, st, ph)
ok = ph[:, 0] < 1e30
c = np.clip(np.floor(ph[ok, :3] * G), 0, G - 1).astype(np.int64)
cnt = np.zeros((G, G, G), np.int64)                      # [z, y, x]
np.add.at(cnt, (c[:, 2], c[:, 1], c[:, 0]), 1)
print("photons in the grid:", ok.sum(), " occupied cells:", (cnt > 0).sum())
cs = np.pad(cnt, 1).cumsum(0).cumsum(1).cumsum(2)
cs = np.pad(cs, ((1, 0), (1, 0), (1, 0)))                # cs[z, y, x] = sum of padded cnt[:z, :y, :x]

def box(z0, z1, y0, y1, x0, x1):                         # sums over padded coords [z0, z1) ...
    return (cs[z1, y1, x1] - cs[z0, y1, x1] - cs[z1, y0, x1] - cs[z1, y1, x0]
            + cs[z0, y0, x1] + cs[z0, y1, x0] + cs[z1, y0, x0] - cs[z0, y0, x0])

for (bx, by, bz) in ((4, 4, 4), (8, 4, 4), (16, 4, 4), (8, 8, 4), (8, 8, 8), (16, 8, 8), (128, 4, 4)):
    zs, ys, xs = np.arange(0, G, bz), np.arange(0, G, by), np.arange(0, G, bx)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    # halo of 1 cell: padded coords shift by +1, so [z0 - 1, z0 + bz + 1) -> [z0, z0 + bz + 2)
    v = box(Z, Z + bz + 2, Y, Y + by + 2, X, X + bx + 2)
    steps = np.ceil(v / 64)
    print(f"brick {bx:3d}x{by}x{bz}: visits {v.sum():9d} = {v.sum() / ok.sum():.2f} per photon; 64-record steps {int(steps.sum()):7d};"
          f" heaviest {v.max():6d}; non-empty {int((v > 0).sum()):6d} of {v.size}")

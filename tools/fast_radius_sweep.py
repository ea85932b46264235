"""The fast bin + gather on config 2's photons over a range of radii (in voxels of the 128^3 light volume): bin and gather times per radius,
for deciding from which box width a photon is better filed once (tiles with a halo + merge) than under every brick it touches.
usage: [CPM_LIB=build/variants/x.so] python tools/fast_radius_sweep.py"""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), S.workspace_tf(), 1024, (128,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
fr.set_planar_records(True)
fr.frame_fast()
torch.cuda.synchronize()


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rv in (0.45, 0.866, 1.2, 1.45, 1.9, 2.4, 2.9, 3.4):
    fr.radius = rv / 128.0
    fr.scale = float(B.relative_irradiance_scale(fr.radius, float(fr.n)))
    fr.brick_table = None; fr.sorted_fast = None          # (sized per radius)
    if not ctx.gather_fast_supported(fr.grid, fr.radius):
        print(f"r = {rv} voxels: not supported"); continue
    fr.bin_fast(); fr.gather_fast()
    tb, tg = timed(fr.bin_fast), timed(fr.gather_fast)
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(20):
        fr.bin_fast(); fr.gather_fast()
    k = ctx.profile_collect(); ctx.profile_enable(False)
    names = "  ".join(f"{kn.split('(')[0][-26:]} {tot / 20 * 1e3:.1f}" for kn, (tot, calls) in sorted(k.items(), key=lambda kv: -kv[1][0])[:4])
    print(f"r = {rv:5.3f} voxels: bin {tb:6.1f} us  gather {tg:6.1f} us  sum {tb + tg:6.1f}   [{names}]   checksum {float(fr.light_volume.double().sum()):.6e}")

"""The sender's pack launch (cpm_bricklist_pack_grid) at config 4's size, rank 1 of 2 under range shards: with the gather's marks and without.
usage: python tools/pack_time.py"""
import importlib, sys
sys.path.insert(0, '.')
import numpy as np, torch
import cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
ctx = B.Context(0)
vdim, (nx, ny), gdim = 512, (2048, 2048), 256
vol = ctx.volume_create(S.heterogeneous_volume(vdim))
lo, hi = sh.shard_range(nx * ny, 1, 2)
fr = P.PhotonFrame(ctx, vol, S.workspace_tf(), (nx, ny), (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0), photon_indices=np.arange(lo, hi, dtype=np.int64))
fr.set_planar_records(True)
nb = 64 ** 3
print(sh.choose_sender_gather(fr, nb))
seg, keep = sh.scratch_segment(torch, ctx.device, nb, 1)
marks = torch.zeros(nb + 16, dtype=torch.uint8, device=ctx.device)
dense = torch.empty_like(fr.light_volume)
fr.trace(); fr.bin_fast(); fr.gather_fast(out=dense, nonzero_bricks=marks)
for name, m in (("with the gather's marks", marks), ("without marks (a pass over the volume)", None)):
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(20):
        ctx.debug_pack_grid_segment(seg, fr.grid, dense, m)
    k = ctx.profile_collect(); ctx.profile_enable(False)
    print(name, {kk[:50]: (round(tot / calls * 1e3, 1), calls) for kk, (tot, calls) in k.items()}, "listed bricks", int(marks[:nb].sum().item()))

"""Gather time on config 2's photons for larger radii and each kernel family (tuning aid)."""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), S.workspace_tf(), 1024, (128,)*3, light_travel_direction=(0.3, 0.5, -1.0))
for _ in range(3): fr.frame()
torch.cuda.synchronize()
for rcells in (0.866, 1.0, 1.2, 1.45, 1.732):
    fr.radius = rcells / 128.0
    ref = None
    for name, force, coop in (("tuned coop", 0, 1), ("tuned 1 wave/brick", 0, 0), ("generic record-major", 2, 1), ("voxel-major", 1, 1)):
        ctx.lib.cpm_debug_force_voxel_gather(ctx.h, force); ctx.lib.cpm_debug_set_gather_coop(ctx.h, coop)
        for _ in range(3): fr.gather()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fr.gather()
        e1.record(); torch.cuda.synchronize()
        lv = fr.light_volume.clone()
        if ref is None: ref = lv
        print(f"r = {rcells} cells: {name:22s} {e0.elapsed_time(e1) / 10 * 1000:8.1f} us  same={bool(torch.equal(lv.view(torch.int32), ref.view(torch.int32)))}")
ctx.lib.cpm_debug_force_voxel_gather(ctx.h, 0); ctx.lib.cpm_debug_set_gather_coop(ctx.h, 1)

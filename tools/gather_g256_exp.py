import sys
sys.path.insert(0, '/root/repo')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), S.workspace_tf(), 1024, (256,)*3, light_travel_direction=(0.3, 0.5, -1.0))
for _ in range(3): fr.frame()
torch.cuda.synchronize()
for name, force, coop in (("tuned coop4", 0, 4), ("tuned 1 wave/brick", 0, 0), ("voxel-major", 1, 1)):
    ctx.lib.cpm_debug_force_voxel_gather(ctx.h, force); ctx.lib.cpm_debug_set_gather_coop(ctx.h, coop)
    for _ in range(3): fr.gather()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fr.gather()
    e1.record(); torch.cuda.synchronize()
    print(f"grid 256^3 r=1.732 cells: {name:22s} {e0.elapsed_time(e1) / 10 * 1000:8.1f} us")

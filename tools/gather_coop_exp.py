"""Gather time for the kernel variants (tuning aid): coop 0 = one wave per brick, 2 / 4 / 8 = waves sharing bricks."""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
for vd, ns, gd in ((256, 1024, 128), (512, 2048, 256)):
    fr = P.PhotonFrame(ctx, S.heterogeneous_volume(vd), S.workspace_tf(), ns, (gd,)*3, light_travel_direction=(0.3, 0.5, -1.0))
    for _ in range(3): fr.frame()
    torch.cuda.synchronize()
    ref = None
    for mode in (0, 4, 1):
        ctx.lib.cpm_debug_set_gather_coop(ctx.h, mode)
        for _ in range(5): fr.gather()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fr.gather()
        e1.record(); torch.cuda.synchronize()
        lv = fr.light_volume.clone()
        if ref is None: ref = lv
        print(f"volume {vd}^3 grid {gd}^3: coop {mode}: {e0.elapsed_time(e1) / 30 * 1000:.1f} us  same={bool(torch.equal(lv.view(torch.int32), ref.view(torch.int32)))}")
    ctx.lib.cpm_debug_set_gather_coop(ctx.h, 1)
    del fr

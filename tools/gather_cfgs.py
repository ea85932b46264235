"""Gather / frame time for the light-volume options of the workspace (volumeSizeOption 1, 1/2, 1/4; 1 and 4 channels)."""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
vol, tf = S.heterogeneous_volume(256), S.workspace_tf()
for gd, ch in ((256, 1), (128, 1), (64, 1), (128, 4)):
    fr = P.PhotonFrame(ctx, vol, tf, 1024, (gd,)*3, light_travel_direction=(0.3, 0.5, -1.0), channels=ch)
    for _ in range(3): fr.frame()
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(20): fr.bin()
    e1.record()
    for _ in range(20): fr.gather()
    e2.record(); torch.cuda.synchronize()
    print(f"grid {gd}^3 x{ch}: bin {e0.elapsed_time(e1) / 20 * 1000:.1f} us, gather {e1.elapsed_time(e2) / 20 * 1000:.1f} us")
    del fr

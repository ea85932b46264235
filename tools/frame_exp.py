"""Frame time on config 2 against the radix tile size (tuning aid)."""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), S.workspace_tf(), 1024, (128,)*3, light_travel_direction=(0.3, 0.5, -1.0))
for _ in range(3): fr.frame()
torch.cuda.synchronize()
for items in (4, 8, 16, 0):
    ctx.lib.cpm_debug_set_sort_items(ctx.h, items)
    for _ in range(10): fr.frame()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): fr.bin()
    e1.record(); torch.cuda.synchronize()
    tb = e0.elapsed_time(e1) / 100 * 1000
    e0.record()
    for _ in range(100): fr.frame()
    e1.record(); torch.cuda.synchronize()
    print(f"items {items}: bin {tb:.1f} us, frame {e0.elapsed_time(e1) / 100 * 1000:.1f} us")

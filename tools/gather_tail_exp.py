"""Experiment: how much of the brick gather is the second round of its 525 work items over 512 workgroups (config 2)?"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), S.workspace_tf(), 1024, (128,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
fr.frame_fast(); torch.cuda.synchronize()
t = fr.brick_table
nb = (t.numel() - 5) // 2
n_items = int(t[nb + 3].item())
def timeit(fn, reps=300):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def frame_with(n):
    # the whole frame, with the table's item count overwritten between bin and gather (wrong volume for n < n_items: timing only)
    def f():
        fr.trace(); fr.bin_fast(); t[nb + 3] = n; fr.gather_fast()
    return f
print(f"{n_items} items")
for n in (n_items, 512, 256, 0):
    t[nb + 3] = n
    print(f"  items = {n:4d}: gather alone {timeit(fr.gather_fast):6.2f} us")

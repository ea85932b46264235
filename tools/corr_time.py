#!/usr/bin/env python3
"""Per-kernel HIP-event times of the fused correlated update (Python driver) at config 3 (TF edit) and config 5 (time steps).
usage: tools/corr_time.py [reps]   (CPM_LIB=build/variants/x.so selects another build of the library)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
import os
import os
ctx = B.Context(0)
vol = S.heterogeneous_volume(256)
base = list(S.WORKSPACE_TF_POINTS); edit = list(base); edit[3] = (0.26,) + base[3][1:]
cm = P.CorrelatedPhotonMapper(ctx, vol, S.workspace_tf(), 1024, (128,) * 3, light_travel_direction=(0.3, 0.5, -1.0), tf_points=base)
cm.full_frame()
def run(n, prof):
    ctx.profile_reset(); ctx.profile_enable(prof)
    ts = []
    for rep in range(n):
        pts = edit if rep % 2 == 0 else base
        torch.cuda.synchronize(); t = time.perf_counter()
        cm.set_transfer_function(pts)
        k = cm.correlated_update()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    kern = ctx.profile_collect() if prof else {}
    ctx.profile_enable(False)
    return ts, kern, k
run(6, False)
ts, _, k = run(reps, False)
print(f"config3 update (python wall): median {np.median(ts):.4f} ms, re-traced {k}")
_, kern, _ = run(reps, True)
tot = 0
for name, (ms, calls) in sorted(kern.items(), key=lambda kv: -kv[1][0]):
    print(f"  {name[:60]:60s} {ms / calls * 1e3:8.2f} us x {calls / reps:.1f}")
    tot += ms / reps
print(f"  sum {tot * 1e3:.1f} us per update")
counter = torch.zeros(1, dtype=torch.int64, device="cuda")
ctx.set_step_counter(counter)
cm.set_transfer_function(edit if reps % 2 == 0 else base)
k = cm.correlated_update()
torch.cuda.synchronize()
ctx.set_step_counter(None)
print(f"  re-trace of {k} photons: {int(counter.item())} Woodcock steps = {int(counter.item()) / max(k, 1):.1f} per photon")

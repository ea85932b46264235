#!/usr/bin/env python3
"""Per-kernel times of the config-2 frame at max_interactions = I (multiple scattering, Henyey-Greenstein g = 0.3).
usage: tools/i4_time.py [I] [frames]"""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
I = int(sys.argv[1]) if len(sys.argv) > 1 else 4
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), S.workspace_tf(), 1024, (128,) * 3, light_travel_direction=(0.3, 0.5, -1.0),
                   max_interactions=I, material=(0.3, 0.0, 0.0, 0.0))
for _ in range(5):
    fr.frame_fast()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(frames):
    fr.frame_fast()
e1.record(); torch.cuda.synchronize()
stored = int((fr.photons[:, 0] != 3.402823466e+38).sum().item())
counter = torch.zeros(1, dtype=torch.int64, device="cuda")
ctx.set_step_counter(counter); fr.trace(); torch.cuda.synchronize(); ctx.set_step_counter(None)
print(f"I = {I}: frame {e0.elapsed_time(e1) / frames * 1e3:8.1f} us, {stored} records stored of {fr.n * I}, {int(counter.item())} Woodcock steps")
ctx.profile_reset(); ctx.profile_enable(True)
for _ in range(frames):
    fr.frame_fast()
k = ctx.profile_collect(); ctx.profile_enable(False)
for kn, (tot, calls) in sorted(k.items(), key=lambda kv: -kv[1][0]):
    print(f"    {kn:48s} {tot / frames * 1e3:8.1f} us/frame  {calls / frames:4.1f} launches")

"""Trace launch time against the number of samples (config 2's volume / TF / light): how much of a launch is fixed.
usage: python tools/trace_size_exp.py"""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
vol = ctx.volume_create(S.heterogeneous_volume(256))
for ny in (64, 128, 256, 512, 1024, 2048):
    fr = P.PhotonFrame(ctx, vol, S.workspace_tf(), (1024, ny), (128,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
    for _ in range(8):
        fr.trace()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fr.trace()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    print(f"1024 x {ny:5d} = {fr.n:8d} samples ({fr.n // 256:6d} workgroups): trace {us:7.2f} us  {fr.n / us / 1e3:6.2f} Gphotons/s")
    del fr

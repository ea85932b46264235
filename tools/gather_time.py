"""Gather kernel time on config 2's photons (current build, default dispatch), light volume compared with the first.
usage (GPU box): python tools/gather_time.py [repeats]"""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), S.workspace_tf(), 1024, (128,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
fr.frame()
ref = fr.light_volume.clone()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for _ in range(5):
        fr.gather()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fr.gather()
    e1.record(); torch.cuda.synchronize()
    print(f"gather {e0.elapsed_time(e1) / 100 * 1e3:7.1f} us  identical={bool(torch.equal(fr.light_volume, ref))}")

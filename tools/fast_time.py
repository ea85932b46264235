"""Per-kernel times (library HIP-event hook) of the exact and the tolerance-mode frame on one workload.
usage (GPU box): python tools/fast_time.py [config2|config4] [frames]"""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 50
vdim, nside, gdim = {"config2": (256, 1024, 128), "config4": (512, 2048, 256), "config1": (64, 256, 32)}[wl]
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(vdim), S.workspace_tf(), nside, (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
exact = fr.frame().clone()
fast = fr.frame_fast().clone()
err = (fast - exact).abs().max().item() / exact.abs().max().item()
print(f"{wl}: max |fast - exact| / max = {err:.3e}; table items = {int(fr.brick_table[fr.brick_table.numel() and 0].item())}")
for name, fn in (("exact", fr.frame), ("fast", fr.frame_fast)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(frames):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:6s} frame {e0.elapsed_time(e1) / frames * 1e3:8.1f} us")
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(frames):
        fn()
    k = ctx.profile_collect(); ctx.profile_enable(False)
    for kn, (tot, calls) in sorted(k.items(), key=lambda kv: -kv[1][0]):
        print(f"    {kn:48s} {tot / frames * 1e3:8.1f} us/frame  {calls / frames:4.1f} launches")

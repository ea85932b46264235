"""Frame and per-kernel times of the fast frame with float8 records and with the two-plane layout (CPM_TRACE_PHOTONS_PLANAR).
usage (GPU box): python tools/planar_time.py [config2|config4] [frames]"""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
vdim, nside, gdim = {"config2": (256, 1024, 128), "config4": (512, 2048, 256), "config1": (64, 256, 32)}[wl]
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(vdim), S.workspace_tf(), nside, (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
for rounds in range(2):
    for planar in (False, True):
        fr.set_planar_records(planar)
        for _ in range(20):
            fr.frame_fast()
        torch.cuda.synchronize()
        best = []
        for rep in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(frames):
                fr.frame_fast()
            e1.record(); torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) / frames * 1e3)
        best.sort()
        print(f"{wl} {'planar ' if planar else 'float8 '} frame median {best[3]:7.2f} us  (min {best[0]:.2f}, max {best[-1]:.2f})")
        if rounds == 1:
            ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(frames):
                fr.frame_fast()
            k = ctx.profile_collect(); ctx.profile_enable(False)
            for kn, (tot, calls) in sorted(k.items(), key=lambda kv: -kv[1][0]):
                print(f"    {kn:48s} {tot / frames * 1e3:8.2f} us/frame")

"""How much do S independent frames in flight (S streams, S contexts) raise throughput?"""
import sys, time
sys.path.insert(0, '.')
import torch, cpm_amd
S_, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
vol_np, tf = S_.heterogeneous_volume(256), S_.workspace_tf()
for S in (1, 2, 3, 4, 6):
    ctxs = [B.Context(0) for _ in range(S)]
    frames = [P.PhotonFrame(c, vol_np, tf, 1024, (128,)*3, light_travel_direction=(0.3, 0.5, -1.0)) for c in ctxs]
    streams = [torch.cuda.Stream() for _ in range(S)]
    for f, s in zip(frames, streams):
        with torch.cuda.stream(s):
            f.frame()
    torch.cuda.synchronize()
    K = 60
    t = time.perf_counter()
    for k in range(K):
        for f, s in zip(frames, streams):
            with torch.cuda.stream(s):
                f.frame()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"streams {S}: {dt / (K * S) * 1e3:.4f} ms per frame, {K * S * 1048576 / dt / 1e6:.0f} Mphotons/s")
    del frames, ctxs

"""Frames in flight on several streams (one context each) against the single-stream result: which differ, by how much."""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
vol, tf = S.heterogeneous_volume(256), S.workspace_tf()
kw = dict(light_travel_direction=(0.3, 0.5, -1.0))
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, vol, tf, 1024, (128,) * 3, **kw)
ref = fr.frame_fast().clone()
print("single stream repeat identical:", bool(torch.equal(fr.frame_fast(), ref)))
NS = 4
ctxs = [B.Context(0) for _ in range(NS)]
frames = [P.PhotonFrame(c, vol, tf, 1024, (128,) * 3, **kw) for c in ctxs]
streams = [torch.cuda.Stream() for _ in range(NS)]
for rnd in range(6):
    for f, st in zip(frames, streams):
        with torch.cuda.stream(st):
            f.trace(); f.bin_fast(); f.gather_fast()
    torch.cuda.synchronize()
    for i, f in enumerate(frames):
        same_ph = bool(torch.equal(f.photons, fr.photons))
        d = (f.light_volume - ref).abs()
        print(f"round {rnd} frame {i}: photons identical={same_ph} light volume identical={bool(torch.equal(f.light_volume, ref))} "
              f"max diff {d.max().item():.3e} n diff {(d > 0).sum().item()} items {int(f.brick_table[4096 + 2].item())} maxpow bits {int(f.brick_table[4097].item())}")

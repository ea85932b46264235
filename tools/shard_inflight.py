"""A rank's frames with 1, 2 and 3 of them in flight (each on its own stream, with its own context and buffers) at config 4's size under a
given sharding -- the segment form (cpm_gather_fast_segment) and the dense form.  usage: python tools/shard_inflight.py [world] [rank] [tiles|range]"""
import importlib
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import torch
import cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = sys.argv[3] if len(sys.argv) > 3 else "range"
vdim, (nx, ny), gdim = 512, (2048, 2048), 256
n_total = nx * ny
if kind == "tiles":
    shard = sh.shard_tiles(n_total, rank, world)
else:
    lo, hi = sh.shard_range(n_total, rank, world)
    shard = np.arange(lo, hi, dtype=np.int64)
vol_np, tf = S.heterogeneous_volume(vdim), S.workspace_tf()
nb = ((gdim + 3) // 4) ** 3
room = (nb + 63) & ~63
K = 3
ctxs = [B.Context(0) for _ in range(K)]
vols = [c.volume_create(vol_np) for c in ctxs]
frames = [P.PhotonFrame(c, v, tf, (nx, ny), (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0), photon_indices=shard) for c, v in zip(ctxs, vols)]
streams = [torch.cuda.Stream() for _ in range(K)]
segs, keep = [], []
for f in frames:
    f.set_planar_records(True)
    buf = torch.empty(16 + room * 272, dtype=torch.uint8, device="cuda")
    ctl = torch.zeros(2, dtype=torch.int32, device="cuda")
    mail = torch.zeros(1, dtype=torch.int64, device="cuda")
    segs.append(B.BricklistSegment(buf.data_ptr(), 8192, room, 7, 1, ctl.data_ptr(), mail.data_ptr()))
    keep.append((buf, ctl, mail))
for form in ("segment", "dense"):
    for inflight in (1, 2, 3):
        def run(reps):
            for it in range(reps):
                j = it % inflight
                with torch.cuda.stream(streams[j]):
                    f = frames[j]
                    f.trace(); f.bin_fast()
                    if form == "segment":
                        f.gather_fast_segment(segs[j])
                    else:
                        f.gather_fast()
        run(12)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t = time.perf_counter()
            run(60)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / 60 * 1e6)
        print(f"N={world} rank {rank} {kind}: {form:8s} {inflight} in flight: {best:.1f} us per frame", flush=True)

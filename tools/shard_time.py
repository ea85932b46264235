#!/usr/bin/env python3
"""One rank's frame (trace + brick bin + brick gather) under weak scaling, measured on ONE GPU with the shard rank r of N would
own: 4096-sample lattice tiles dealt round-robin (sharding.shard_tiles, bench.py's default) against contiguous ranges (slabs
of the light plane).  usage: tools/shard_time.py [frames]"""
import sys
sys.path.insert(0, '.')
import importlib
import numpy as np, torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = B.Context(0)
vol = ctx.volume_create(S.heterogeneous_volume(256))
tf = S.workspace_tf()
for world in (1, 2, 4, 8):
    lattice = (1024, 1024 * world)
    n_total = lattice[0] * lattice[1]
    for mode in ("tiles", "range"):
        if world == 1 and mode == "range":
            continue
        for rank in sorted({0, world // 2, world - 1}):
            if mode == "tiles":
                idx = sh.shard_tiles(n_total, rank, world)
            else:
                lo, hi = sh.shard_range(n_total, rank, world)
                idx = np.arange(lo, hi, dtype=np.int64)
            fr = P.PhotonFrame(ctx, vol, tf, lattice, (128,) * 3, light_travel_direction=(0.3, 0.5, -1.0), photon_indices=idx)
            for _ in range(5):
                fr.frame_fast()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(frames):
                fr.frame_fast()
            e1.record(); torch.cuda.synchronize()
            ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(frames):
                fr.frame_fast()
            k = ctx.profile_collect(); ctx.profile_enable(False)
            per = {kn.split("<")[0]: tot / frames * 1e3 for kn, (tot, calls) in k.items()}
            print(f"N={world} {mode:5s} rank {rank}: frame {e0.elapsed_time(e1) / frames * 1e3:6.1f} us | " +
                  " ".join(f"{kn.replace('fast_', '').replace('_kernel', '')} {v:5.1f}" for kn, v in sorted(per.items())), flush=True)
            del fr

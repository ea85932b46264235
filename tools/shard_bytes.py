"""What the frame's exchange would put on xGMI for N = 2 / 4 / 8 ranks under both shardings -- measured on ONE GPU: every rank's frame is
run in turn with the shard that rank would own, the non-zero 4x4x4 bricks of its light volume are counted (the gather's own marks), and
the three exchanges' bytes per link follow from those counts (sharding.exchange_model: dense ring reduce, union-of-bricks reduce, per-rank
brick lists to the root).  Also the per-rank frame time with that shard (the compute the exchange must hide behind).

usage (GPU box): python tools/shard_bytes.py [config2|config4] [out.json]     config2: weak scaling (1 048 576 photons per rank),
                                                                               config4: strong (4 194 304 photons in all)"""
import importlib
import json
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
out_path = sys.argv[2] if len(sys.argv) > 2 else None
vdim, (nx, ny), gdim, scaling = {"config2": (256, (1024, 1024), 128, "weak"), "config4": (512, (2048, 2048), 256, "strong")}[wl]
ctx = B.Context(0)
vol = ctx.volume_create(S.heterogeneous_volume(vdim))
tf = S.workspace_tf()
nb = ((gdim + 3) // 4) ** 3
report = {"workload": wl, "scaling": scaling, "light_volume": [gdim] * 3, "n_bricks_4x4x4": nb, "dense_bytes": gdim ** 3 * 4,
          "method": "one GPU, every rank's frame in turn with its shard; bricks from cpm_gather_fast_marked; bytes per link from sharding.exchange_model "
                    "(latency 30 us per collective, 100 GB/s per link: arithmetic, not a measurement over xGMI)", "runs": []}


def frame_time(fr, reps=30):
    for _ in range(5):
        fr.frame_fast()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fr.frame_fast()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for world in (2, 4, 8):
    lattice, n_total = ((nx, ny * world), nx * ny * world) if scaling == "weak" else ((nx, ny), nx * ny)
    for kind in ("tiles", "range"):
        union = torch.zeros(nb, dtype=torch.uint8, device=ctx.device)
        counts, times = [], []
        ranks = range(world) if world <= 4 else (0, 3, 7)     # (8 ranks: first, a middle and the last -- the others lie between)
        for r in ranks:
            if kind == "tiles":
                shard = sh.shard_tiles(n_total, r, world)
            else:
                lo, hi = sh.shard_range(n_total, r, world)
                shard = np.arange(lo, hi, dtype=np.int64)
            fr = P.PhotonFrame(ctx, vol, tf, lattice, (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0), photon_indices=shard)
            fr.set_planar_records(True)
            marks = torch.zeros(nb + 16, dtype=torch.uint8, device=ctx.device)
            fr.trace(); fr.bin_fast(); fr.gather_fast(nonzero_bricks=marks)
            torch.cuda.synchronize()
            counts.append(int(marks[:nb].sum().item()))
            union |= marks[:nb]
            times.append(round(frame_time(fr), 1))
            del fr
        n_union = int(union.sum().item())
        if world > 4 and kind == "range":   # slabs are disjoint up to their borders: the union of all 8 is about the sum of the slabs
            n_union = min(nb, int(np.mean(counts) * world))
        model = sh.exchange_model(nb, 1, world, n_union, max(counts), gdim ** 3)
        row = {"ranks": world, "shards": kind, "ranks_measured": list(ranks), "lit_bricks_per_rank": counts, "union_bricks": n_union,
               "frame_us_per_rank": times, "exchange": model}
        report["runs"].append(row)
        print(f"{wl} N={world} {kind:6s} lit/rank {counts} union {n_union} frame us {times}  "
              + "  ".join(f"{k}: {v['bytes_per_link'] / 1e6:.2f} MB ~{v['model_us']:.0f} us" for k, v in model.items()), flush=True)
if out_path:
    json.dump(report, open(out_path, "w"), indent=1)

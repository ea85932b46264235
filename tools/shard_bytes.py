"""What the frame's exchange would put on xGMI for N = 2 / 4 / 8 ranks under both shardings, and what a rank's frame costs in the form it
would run -- measured on ONE GPU: every rank's frame is run in turn with the shard that rank would own.

Per rank: the non-zero 4x4x4 bricks of its light volume (the gather's own marks); the frame with a DENSE light volume (trace, bin,
cpm_gather_fast_marked: what the display GPU runs, and what every rank ran until round 5) and the frame of a rank that is NOT the display
GPU (trace, bin, cpm_gather_fast_segment: the non-zero bricks straight into its brick-list segment -- no dense volume, no zeros, no list /
pack pass).  Per (N, shards): the root's post-receive work -- its two launches (brick -> slot tables, sum in rank order) over the N - 1
segments those frames really produced, timed with events through the measurement hook (cpm_debug_root_add_segments).  The three exchanges'
bytes per link follow from the counts (sharding.exchange_model: dense ring reduce, union-of-bricks reduce, per-rank brick lists).

usage (GPU box): python tools/shard_bytes.py [config2|config4] [out.json]     config2: weak scaling (1 048 576 photons per rank),
                                                                               config4: strong (4 194 304 photons in all)"""
import ctypes as C
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
lib = B.load_library()
lib.cpm_debug_root_add_segments.restype = C.c_int
lib.cpm_debug_root_add_segments.argtypes = [C.c_void_p, C.POINTER(B.BricklistSegment), C.c_int, C.POINTER(B.GridDesc), C.c_void_p, C.c_void_p, C.c_void_p]
vol = ctx.volume_create(S.heterogeneous_volume(vdim))
tf = S.workspace_tf()
nb = ((gdim + 3) // 4) ** 3
room = (nb + 63) & ~63
report = {"workload": wl, "scaling": scaling, "light_volume": [gdim] * 3, "n_bricks_4x4x4": nb, "dense_bytes": gdim ** 3 * 4,
          "method": "one GPU, every rank's frame in turn with its shard; frame_us_dense = trace + bin + cpm_gather_fast_marked (a dense light volume: "
                    "the display GPU's frame), frame_us_segment = trace + bin + cpm_gather_fast_segment (a rank that is not the display GPU); "
                    "root_add_us = the root's two post-receive launches over the N - 1 segments those frames produced (events, median of 20); bytes per "
                    "link from sharding.exchange_model (assumed constants: arithmetic, not a measurement over xGMI)", "runs": []}


def timed_us(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def make_segment(ticket):
    buf = torch.empty(16 + room * 272, dtype=torch.uint8, device=ctx.device)
    ctl = torch.zeros(2, dtype=torch.int32, device=ctx.device)
    mail = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    return B.BricklistSegment(buf.data_ptr(), room, room, ticket, 1, ctl.data_ptr(), mail.data_ptr()), (buf, ctl, mail)


for world in (2, 4, 8):
    lattice, n_total = ((nx, ny * world), nx * ny * world) if scaling == "weak" else ((nx, ny), nx * ny)
    for kind in ("tiles", "range"):
        union = torch.zeros(nb, dtype=torch.uint8, device=ctx.device)
        counts, t_dense, t_seg, segs, keep = [], [], [], [], []
        grid_desc = None
        for r in range(world):
            if kind == "tiles":
                shard = sh.shard_tiles(n_total, r, world)
            else:
                lo, hi = sh.shard_range(n_total, r, world)
                shard = np.arange(lo, hi, dtype=np.int64)
            fr = P.PhotonFrame(ctx, vol, tf, lattice, (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0), photon_indices=shard)
            fr.set_planar_records(True)
            grid_desc = fr.grid
            marks = torch.zeros(nb + 16, dtype=torch.uint8, device=ctx.device)
            fr.trace(); fr.bin_fast(); fr.gather_fast(nonzero_bricks=marks)
            torch.cuda.synchronize()
            counts.append(int(marks[:nb].sum().item()))
            union |= marks[:nb]
            t_dense.append(round(timed_us(lambda: (fr.trace(), fr.bin_fast(), fr.gather_fast(nonzero_bricks=marks))), 1))
            seg, bufs = make_segment(7)
            # (the capacity the exchange would have settled on: this rank's count * 1.25 + 64 -- what the root's launches are sized by)
            seg.capacity = sh.bricklist_capacity(nb, counts[-1])
            t_seg.append(round(timed_us(lambda: (fr.trace(), fr.bin_fast(), fr.gather_fast_segment(seg))), 1))
            if r > 0:
                segs.append(seg); keep.append(bufs)
            del fr
        n_union = int(union.sum().item())
        # the root's work behind the receive: N - 1 real segments into a dense volume (any: the adds' cost does not depend on its values)
        total = torch.zeros(gdim ** 3, dtype=torch.float32, device=ctx.device)
        slot_of = torch.zeros(world * nb, dtype=torch.int32, device=ctx.device)   # (N - 1 brick -> slot tables + the who-lists-what words)
        arr = (B.BricklistSegment * (world - 1))(*segs)

        def root_add():
            rc = lib.cpm_debug_root_add_segments(ctx.h, arr, world - 1, C.byref(grid_desc), ctx._ptr(total), ctx._ptr(slot_of), ctx._stream())
            assert rc == 0
        samples = []
        for _ in range(20):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); root_add(); e1.record(); torch.cuda.synchronize()
            samples.append(e0.elapsed_time(e1) * 1e3)
        root_us = round(float(np.median(samples[5:])), 1)
        root_b2b = round(timed_us(root_add, reps=20), 1)
        model = sh.exchange_model(nb, 1, world, n_union, max(counts[1:]), gdim ** 3)
        row = {"ranks": world, "shards": kind, "lit_bricks_per_rank": counts, "union_bricks": n_union,
               "frame_us_dense": t_dense, "frame_us_segment": t_seg, "root_add_us": root_us, "root_add_us_back_to_back": root_b2b,
               "root_add_launches": 2, "listed_bricks_at_root": int(sum(counts[1:])), "exchange": model}
        report["runs"].append(row)
        print(f"{wl} N={world} {kind:6s} lit/rank {counts} union {n_union} dense frame us {t_dense} segment frame us {t_seg} root add us {root_us} "
              f"({root_b2b} back to back)  " + "  ".join(f"{k}: {v['bytes_per_link'] / 1e6:.2f} MB ~{v['model_us']:.0f} us" for k, v in model.items() if k != "constants"),
              flush=True)
        del segs, keep, arr, total, slot_of
if out_path:
    json.dump(report, open(out_path, "w"), indent=1)

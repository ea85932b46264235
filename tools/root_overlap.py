"""Does the display GPU's post-receive work (the two launches that add the senders' brick lists) run BESIDE its own next frame, or do the
two serialise?  One GPU, one process: stream A runs the root's frames (trace, bin, dense gather) at config 4's size with the root's shard,
stream B -- behind an event recorded after each gather, as the exchange stream is -- runs the root's two launches over the N - 1 segments
the other ranks' frames produced (measurement hook cpm_debug_root_add_segments; the receive itself is RCCL's and not here).  Run under
`rocprofv3 --kernel-trace`; tools/overlap_from_trace.py turns the kernel start / end stamps into the overlap.
usage: python tools/root_overlap.py [world] [frames]"""
import importlib
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
vdim, (nx, ny), gdim = 512, (2048, 2048), 256
n_total = nx * ny
ctx = B.Context(0)
vol = ctx.volume_create(S.heterogeneous_volume(vdim))
tf = S.workspace_tf()
nb = ((gdim + 3) // 4) ** 3
segs, keep, root, listed = [], [], None, 0
# --root-share f: the display GPU traces f of an equal share, the others split the rest (its adds are part of ITS frame period: a lighter shard evens the ranks out)
root_share = float(sys.argv[sys.argv.index("--root-share") + 1]) if "--root-share" in sys.argv else 1.0
n_root = int(round(n_total / world * root_share))
bounds = [0, n_root] + [n_root + (n_total - n_root) * (r + 1) // (world - 1) for r in range(world - 1)]
for r in range(world):
    lo, hi = bounds[r], bounds[r + 1]
    fr = P.PhotonFrame(ctx, vol, tf, (nx, ny), (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0), photon_indices=np.arange(lo, hi, dtype=np.int64))
    fr.set_planar_records(True)
    if r == 0:
        root = fr
        continue
    marks = torch.zeros(nb + 16, dtype=torch.uint8, device=ctx.device)
    fr.trace(); fr.bin_fast(); fr.gather_fast(nonzero_bricks=marks)
    count = int(marks[:nb].sum().item())
    listed += count
    seg, bufs = sh.scratch_segment(torch, ctx.device, nb, 1, capacity=sh.bricklist_capacity(nb, count), ticket=3)
    fr.gather_fast_segment(seg)
    if "--time" in sys.argv and r in (1, world // 2, world - 1):   # a sender's frame with this shard (segment form), back to back
        for _ in range(10):
            fr.trace(); fr.bin_fast(); fr.gather_fast_segment(seg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            fr.trace(); fr.bin_fast(); fr.gather_fast_segment(seg)
        e1.record(); torch.cuda.synchronize()
        print(f"sender rank {r}: {hi - lo} photons, frame {e0.elapsed_time(e1) / 40 * 1e3:.1f} us (segment form)", flush=True)
    segs.append(seg); keep.append(bufs)
    del fr
torch.cuda.synchronize()
print("photons: root", bounds[1], "others", [bounds[r + 1] - bounds[r] for r in range(1, world)])
slot_of = torch.zeros(world * nb, dtype=torch.int32, device=ctx.device)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
totals = [root.light_volume, torch.empty_like(root.light_volume)]
done = [None, None]
for k in range(frames + 4):
    b = k & 1
    with torch.cuda.stream(sa):
        root.trace(); root.bin_fast()
        if done[b] is not None:
            sa.wait_event(done[b])       # (a buffer is gathered into again only after its adds -- two frames back)
        root.gather_fast(out=totals[b])
        ready = torch.cuda.Event(); ready.record(sa)
    with torch.cuda.stream(sb):
        sb.wait_event(ready)
        ctx.debug_root_add_segments(segs, root.grid, totals[b], slot_of)
        done[b] = torch.cuda.Event(); done[b].record(sb)
torch.cuda.synchronize()
print("frames", frames + 4, "senders", world - 1, "listed bricks", listed)
if "--time" in sys.argv:   # (not under the profiler) the root's frame period with and without the adds beside it, wall clock
    import time

    def loop(with_adds, reps=200):
        done = [None, None]
        for k in range(reps):
            b = k & 1
            with torch.cuda.stream(sa):
                root.trace(); root.bin_fast()
                if done[b] is not None:
                    sa.wait_event(done[b])
                root.gather_fast(out=totals[b])
                ready = torch.cuda.Event(); ready.record(sa)
            if with_adds:
                with torch.cuda.stream(sb):
                    sb.wait_event(ready)
                    ctx.debug_root_add_segments(segs, root.grid, totals[b], slot_of)
                    done[b] = torch.cuda.Event(); done[b].record(sb)
    for with_adds in (False, True, False, True):
        loop(with_adds, 30); torch.cuda.synchronize()
        t = time.perf_counter(); loop(with_adds); torch.cuda.synchronize()
        print(f"root frame period, adds {'beside' if with_adds else 'absent'}: {(time.perf_counter() - t) / 200 * 1e6:.1f} us", flush=True)

"""One rank of the N-rank test of cpm_reduce_grid_bricklists over the RCCL test double (tests/test_fake_rccl_gpu.py starts 2 and 4 of
these on the box's one GPU).  usage: worker_lists.py <rank> <world> <dir> <root>"""
import importlib
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
import cpm_amd  # noqa: E402

rank, world, out, root = int(sys.argv[1]), int(sys.argv[2]), Path(sys.argv[3]), int(sys.argv[4])
B = cpm_amd.binding
sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
torch.zeros(1, device="cuda")
ctx = B.Context(0)


def exchange_id(name):
    f = out / name
    if rank == 0:
        uid = ctx.comm_unique_id()
        tmp = f.with_suffix(".tmp")
        tmp.write_bytes(uid)
        tmp.rename(f)
        return uid
    for _ in range(40000):
        if f.exists():
            break
        time.sleep(0.001)
    return f.read_bytes()


def partial(dims, ch, k, who, n_ranks):
    """Rank `who`'s light volume of frame k: its slab of y rows (a contiguous photon range enters through a slab of the light plane)
    plus one border row of the next rank's; the lit depth grows with k and comes back.  Multiples of 1/8: sums are exact."""
    dx, dy, dz = dims
    g = np.zeros((dz, dy, dx, ch), np.float32)
    rng = np.random.default_rng(4000 * k + 31 * who + dx)
    depth = min(dz, (2 + 10 * k) if k < 4 else 4)
    lit = rng.random((depth, dy, dx)) < 0.35
    rows = np.zeros(dy, bool)
    rows[who * dy // n_ranks:min(dy, (who + 1) * dy // n_ranks + 1)] = True
    lit &= rows[None, :, None]
    vals = rng.integers(1, 1000, (depth, dy, dx, ch)).astype(np.float32) / np.float32(8.0)
    g[:depth] = vals * lit[..., None]
    return g.reshape(-1)


results = {}
comm = ctx.comm_create(exchange_id("uid_lists.bin"), rank, world)
# 1. the entry points themselves, ticket by ticket: with and without the gather's marks, a ragged 4-channel grid too
for dims, ch in (((32, 32, 32), 1), ((20, 13, 9), 4)):
    gd = B.default_grid_desc(dims, ch)
    br = ctx.bricklist_reduce_create(comm, gd, root)
    nb = br.n_bricks
    infos = []
    for k in range(8):
        mine = partial(dims, ch, k, rank, world)
        g = torch.from_numpy(mine).to(ctx.device)
        marks = None
        if k % 2 == 1:   # what cpm_gather_fast_marked would have left
            vox = (mine.reshape(dims[2], dims[1], dims[0], ch) != 0).any(axis=3)
            pz, py, px = (-dims[2]) % 4, (-dims[1]) % 4, (-dims[0]) % 4
            vox = np.pad(vox, ((0, pz), (0, py), (0, px)))
            nzb = vox.reshape(vox.shape[0] // 4, 4, vox.shape[1] // 4, 4, vox.shape[2] // 4, 4).any(axis=(1, 3, 5)).reshape(-1)
            marks = torch.zeros(nb + 16, dtype=torch.uint8, device=ctx.device)
            marks[:nb] = torch.from_numpy(nzb.astype(np.uint8)).to(ctx.device)
        t = br.start(g, nonzero_bricks=marks)
        i = br.complete(t)
        torch.cuda.synchronize()
        results[f"lists_{dims[0]}_{ch}_{k}"] = g.cpu().numpy()
        infos.append((i.n_own, i.capacity, i.resent, i.sent_bytes, i.received_bytes, i.listed_bricks, i.n_bricks, i.dense_bytes))
    results[f"lists_{dims[0]}_{ch}_info"] = np.array(infos, np.int64)
    br.close()
comm.close()


# 2. as bench.py drives it: the double-buffered reducer over the transport, two tickets in flight
class FileIdTransport(sh.RcclTransport):
    def __init__(self, ctx, rank, world, uid_name="uid_lists2.bin"):
        import torch as _t
        self.ctx, self.torch, self.world, self.rank, self.root = ctx, _t, world, rank, root
        self.comm = ctx.comm_create(exchange_id(uid_name), rank, world)
        self.stream = _t.cuda.Stream(device=ctx.device)


tr = FileIdTransport(ctx, rank, world)
dims = (32, 32, 32)
gd = B.default_grid_desc(dims, 1)
red = sh.OverlappedGridReducer(torch.zeros(32 ** 3, device=ctx.device), tr, lists=gd, root=root)
assert red.active and red.lists
for k in range(7):
    buf = red.acquire(k)
    if k >= 2:
        results[f"reducer_{k - 2}"] = buf.cpu().numpy().copy()      # the exchange of frame k - 2 is complete here
    buf.copy_(torch.from_numpy(partial(dims, 1, k, rank, world)).to(ctx.device))
    red.reduce(k)
red.flush()
torch.cuda.synchronize()
results["reducer_5"] = red.result(5).cpu().numpy().copy()
results["reducer_6"] = red.result(6).cpu().numpy().copy()
results["reducer_info"] = np.array([(i["n_own"], i["capacity"], i["resent"], i["sent_bytes"], i["received_bytes"]) for i in red.info], np.int64)
tr.close()

# 3. frames whose senders have NO dense light volume: open -> cpm_gather_fast_segment -> exchange -> complete.  Real photons (64^3 volume,
#    128 x 128 lattice, 32^3 light volume) under both shardings -- tile shards make every rank light every brick: bricks listed by ALL
#    senders, summed at the root in rank order from inexact values.  The dense gather of the same records is kept for the test's sum.
S, P = cpm_amd.synthetic, cpm_amd.pipeline
vol_np, tf = S.heterogeneous_volume(64), S.workspace_tf()
n_total = 128 * 128
for kind in ("tiles", "range"):
    if kind == "tiles":
        shard = sh.shard_tiles(n_total, rank, world, tile=1024)
    else:
        lo, hi = sh.shard_range(n_total, rank, world)
        shard = np.arange(lo, hi, dtype=np.int64)
    fr = P.PhotonFrame(ctx, vol_np, tf, (128, 128), (32, 32, 32), light_travel_direction=(0.3, 0.5, -1.0), photon_indices=shard)
    fr.set_planar_records(True)
    comm3 = ctx.comm_create(exchange_id(f"uid_seg_{kind}.bin"), rank, world)
    br = ctx.bricklist_reduce_create(comm3, fr.grid, root)
    dense = torch.empty_like(fr.light_volume)
    infos = []
    for k in range(5):
        if k == 2:   # a thin medium from here on: the photons reach deep into the volume, every rank's list grows past the capacity its last counts gave it
            fr.tf.update(S.homogeneous_tf(0.02))
        fr.trace(); fr.bin_fast()
        fr.gather_fast(out=dense)
        ticket, seg = br.open()
        assert bool(seg.segment) == (rank != root)
        if seg.segment:
            fr.gather_fast_segment(seg)
        total = dense.clone()
        br.exchange(ticket, total if rank == root else None)
        i = br.complete(ticket)
        torch.cuda.synchronize()
        results[f"seg_{kind}_{k}"] = total.cpu().numpy()
        results[f"seg_{kind}_{k}_dense"] = dense.cpu().numpy()
        infos.append((i.n_own, i.capacity, i.resent, i.sent_bytes, i.received_bytes, i.listed_bricks, i.n_bricks, i.dense_bytes))
    results[f"seg_{kind}_info"] = np.array(infos, np.int64)
    br.close()
    comm3.close()
    del fr

# 4. bench.py's loop in that form: segment_for / gather / reduce with two tickets in flight
tr = FileIdTransport(ctx, rank, world, "uid_lists3.bin")
shard = sh.shard_tiles(n_total, rank, world, tile=1024)
fr = P.PhotonFrame(ctx, vol_np, tf, (128, 128), (32, 32, 32), light_travel_direction=(0.3, 0.5, -1.0), photon_indices=shard)
fr.set_planar_records(True)
red = sh.OverlappedGridReducer(fr.light_volume, tr, lists=fr.grid, root=root)
dense = torch.empty_like(fr.light_volume)
for k in range(6):
    fr.trace(); fr.bin_fast()
    if k == 0:
        fr.gather_fast(out=dense)
        results["loop_dense"] = dense.cpu().numpy()
    buf = red.acquire(k)
    if k >= 2 and rank == root:
        results[f"loop_{k - 2}"] = buf.cpu().numpy().copy()
    seg = red.segment_for(k)
    assert (seg is None) == (rank == root)
    if seg is not None:
        fr.gather_fast_segment(seg)
    else:
        fr.gather_fast(out=buf)
    red.reduce(k)
red.flush()
torch.cuda.synchronize()
if rank == root:
    results["loop_4"] = red.result(4).cpu().numpy().copy()
    results["loop_5"] = red.result(5).cpu().numpy().copy()
results["loop_info"] = np.array([(i["n_own"], i["capacity"], i["resent"], i["sent_bytes"], i["received_bytes"]) for i in red.info], np.int64)
tr.close()

# 5. a seeded walk: ragged grids, both channel counts, lit sets that grow, shrink and jump from ticket to ticket (capacities follow two tickets
#    behind: lists outgrow their segments again and again), INEXACT values, three tickets in flight before the first is completed
def fuzz_partial(dims, ch, k, who):
    dx, dy, dz = dims
    rng = np.random.default_rng(90000 + 1000 * k + 37 * who + dx * dy)
    density = (0.01, 0.3, 0.02, 0.6, 0.6, 0.05, 0.9, 0.001, 0.4, 0.0)[k % 10]
    g = np.zeros((dz, dy, dx, ch), np.float32)
    lit = rng.random((dz, dy, dx)) < density
    if who % 2 == 1:
        lit[: dz // 2] = False                      # (odd ranks light the far half only: bricks with few and with many listers)
    g[lit] = rng.random((int(lit.sum()), ch), dtype=np.float32) + np.float32(0.01)
    return g.reshape(-1)


comm5 = ctx.comm_create(exchange_id("uid_fuzz.bin"), rank, world)
for fi, (dims, ch) in enumerate((((36, 20, 28), 1), ((17, 33, 12), 4), ((64, 16, 16), 1))):
    gd = B.default_grid_desc(dims, ch)
    br = ctx.bricklist_reduce_create(comm5, gd, root)
    infos, pending = [], []
    for k in range(10):
        g = torch.from_numpy(fuzz_partial(dims, ch, k, rank)).to(ctx.device)
        pending.append((k, g, br.start(g)))
        if len(pending) == 3:
            kk, gg, tt = pending.pop(0)
            i = br.complete(tt)
            torch.cuda.synchronize()
            results[f"fuzz_{fi}_{kk}"] = gg.cpu().numpy()
            infos.append((i.n_own, i.capacity, i.resent))
    for kk, gg, tt in pending:
        i = br.complete(tt)
        torch.cuda.synchronize()
        results[f"fuzz_{fi}_{kk}"] = gg.cpu().numpy()
        infos.append((i.n_own, i.capacity, i.resent))
    results[f"fuzz_{fi}_info"] = np.array(infos, np.int64)
    br.close()
comm5.close()
np.savez(out / f"rank{rank}.npz", **results)
print("worker", rank, "done")

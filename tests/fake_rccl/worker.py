"""One rank of the two-rank test of the C-ABI's multi-GPU path over the RCCL test double (tests/test_fake_rccl_gpu.py starts two
of these on the box's one GPU).  usage: worker.py <rank> <world> <dir>"""
import importlib
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
import cpm_amd  # noqa: E402

rank, world, out = int(sys.argv[1]), int(sys.argv[2]), Path(sys.argv[3])
B = cpm_amd.binding
sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
torch.zeros(1, device="cuda")
ctx = B.Context(0)

# the communicator id travels through a file (any channel will do: cpm.h)
idfile = out / "uid.bin"
if rank == 0:
    uid = ctx.comm_unique_id()
    tmp = out / "uid.tmp"
    tmp.write_bytes(uid)
    tmp.rename(idfile)
else:
    for _ in range(20000):
        if idfile.exists():
            break
        time.sleep(0.001)
    uid = idfile.read_bytes()
comm = ctx.comm_create(uid, rank, world)
assert comm.size == world and comm.rank == rank
results = {}


def partial(dims, ch, k, who):
    """Rank `who`'s light volume of frame k: a lit slab whose depth changes with k (grows past a capacity sized two frames
    before, then shrinks), each rank lighting other voxels of it; values are multiples of 1/8 (sums exact in any order)."""
    dx, dy, dz = dims
    g = np.zeros((dz, dy, dx, ch), np.float32)
    rng = np.random.default_rng(1000 * k + 17 * who + dx)
    depth = min(dz, (2 + 7 * k) if k < 5 else 4)
    lit = rng.random((depth, dy, dx)) < 0.3
    vals = rng.integers(1, 1000, (depth, dy, dx, ch)).astype(np.float32) / np.float32(8.0)
    g[:depth] = vals * lit[..., None]
    return g.reshape(-1)


# 1. dense collectives through the C-ABI with two ranks
g = torch.from_numpy(partial((16, 16, 16), 1, 0, rank)).to(ctx.device)
ctx.allreduce_grid(comm, g)
results["dense_allreduce"] = g.cpu().numpy()
g = torch.from_numpy(partial((16, 16, 16), 1, 1, rank)).to(ctx.device)
recv = torch.full_like(g, -1.0)
ctx.reduce_grid(comm, g, recv if rank == 0 else None, 0)
results["dense_reduce_root0"] = recv.cpu().numpy()

# 2. the sparse reduce: frames whose union grows, overflows, goes dense by policy and comes back; in place and into a separate total
for dims, ch in (((32, 32, 32), 1), ((20, 13, 9), 4)):
    gd = B.default_grid_desc(dims, ch)
    sr = ctx.sparse_reduce_create(comm, gd)
    infos = []
    for k in range(8):
        p = torch.from_numpy(partial(dims, ch, k, rank)).to(ctx.device)
        if k % 2 == 0:
            t = sr.start(p)                                  # in place
            i = sr.complete(t)
            res = p
        else:
            total = torch.full_like(p, 7.0)
            t = sr.start(p, total)                           # separate total: zeros outside the union
            i = sr.complete(t)
            res = total
        torch.cuda.synchronize()
        results[f"sparse_{dims[0]}_{ch}_{k}"] = res.cpu().numpy()
        infos.append((i.n_union, i.capacity, i.mode, i.reduce_bytes))
    results[f"sparse_{dims[0]}_{ch}_info"] = np.array(infos, np.int64)
    # root reduce: only rank 1 receives
    p = torch.from_numpy(partial(dims, ch, 2, rank)).to(ctx.device)
    i = sr.complete(sr.start(p, root=1))
    torch.cuda.synchronize()
    results[f"sparse_{dims[0]}_{ch}_root1"] = p.cpu().numpy()
    # delta path: a touched-brick mask per rank, a separate total that keeps everything else
    nb = sr.n_bricks
    rng = np.random.default_rng(5 + rank)
    mask = (rng.random(nb) < 0.05).astype(np.uint8)
    p = torch.from_numpy(partial(dims, ch, 3, rank)).to(ctx.device)
    total = torch.full_like(p, -2.0)
    i = sr.complete(sr.start(p, total, brick_mask=torch.from_numpy(mask).to(ctx.device), capacity=max(nb // 2, 1)))
    torch.cuda.synchronize()
    results[f"sparse_{dims[0]}_{ch}_delta"] = total.cpu().numpy()
    results[f"sparse_{dims[0]}_{ch}_delta_mask"] = mask
    results[f"sparse_{dims[0]}_{ch}_delta_union"] = np.array([i.n_union, i.mode], np.int64)
    # a MIXED frame: rank 0 rebuilds its volume and hands in every brick it had lit or lights now (two masks OR-ed on the
    # device: cpm_brick_mask_or), rank 1 updates a few bricks and hands in those -- both as TOUCHED masks, into the standing sum
    dx, dy, dz = dims
    bxn, byn = (dx + 3) // 4, (dy + 3) // 4
    zz, yy, xx = np.meshgrid(np.arange(dz), np.arange(dy), np.arange(dx), indexing="ij")
    brick_of = np.repeat(((xx // 4) + bxn * ((yy // 4) + byn * (zz // 4))).reshape(-1), ch)

    def nz_mask(g):
        m = np.zeros(nb, np.uint8)
        m[np.unique(brick_of[g != 0])] = 1
        return m
    standing = [partial(dims, ch, 4, w) for w in (0, 1)]
    total = torch.from_numpy(standing[0] + standing[1]).to(ctx.device)
    if rank == 0:
        new = partial(dims, ch, 6, 0)
        m_dev = torch.from_numpy(nz_mask(standing[0])).to(ctx.device)
        ctx.brick_mask_or(m_dev, torch.from_numpy(nz_mask(new)).to(ctx.device))
    else:
        t1 = (np.random.default_rng(77).random(nb) < 0.04).astype(np.uint8)
        new = standing[1].copy()
        sel = t1[brick_of].astype(bool)
        new[sel] = partial(dims, ch, 7, 1)[sel]
        m_dev = torch.from_numpy(t1).to(ctx.device)
    p = torch.from_numpy(new).to(ctx.device)
    i = sr.complete(sr.start(p, total, brick_mask=m_dev))
    torch.cuda.synchronize()
    results[f"sparse_{dims[0]}_{ch}_mixed"] = total.cpu().numpy()
    results[f"sparse_{dims[0]}_{ch}_mixed_new"] = new
    sr.close()
comm.close()

# 3. the sharding layer's transport + the double-buffered reducer, as bench.py drives them (marks from the "gather" on odd frames)
os.environ["CPM_FAKE_UID_FILE"] = str(out / "uid2.bin")


class FileIdTransport(sh.RcclTransport):
    """RcclTransport whose communicator id travels through a file instead of torch.distributed (none here)."""

    def __init__(self, ctx, rank, world):
        import torch as _t
        self.ctx, self.torch, self.world, self.rank, self.root = ctx, _t, world, rank, None
        f = Path(os.environ["CPM_FAKE_UID_FILE"])
        if rank == 0:
            uid2 = ctx.comm_unique_id()
            tmp2 = f.with_suffix(".tmp")
            tmp2.write_bytes(uid2)
            tmp2.rename(f)
        else:
            for _ in range(20000):
                if f.exists():
                    break
                time.sleep(0.001)
            uid2 = f.read_bytes()
        self.comm = ctx.comm_create(uid2, rank, world)
        self.stream = _t.cuda.Stream(device=ctx.device)


tr = FileIdTransport(ctx, rank, world)
dims = (32, 32, 32)
gd = B.default_grid_desc(dims, 1)
first = torch.zeros(32 ** 3, device=ctx.device)
red = sh.OverlappedGridReducer(first, tr, sparse=gd)
assert red.active and red.sparse
bx = 8
for k in range(7):
    buf = red.acquire(k)
    if k >= 2:
        results[f"reducer_{k - 2}"] = buf.cpu().numpy().copy()      # the reduce of frame k - 2 is complete here
    mine = partial(dims, 1, k, rank)
    buf.copy_(torch.from_numpy(mine).to(ctx.device))
    marked = k % 2 == 1
    if marked:
        m = red.marks_for(k)
        vox = (mine.reshape(32, 32, 32) != 0)
        nzb = vox.reshape(8, 4, 8, 4, 8, 4).any(axis=(1, 3, 5)).reshape(-1)
        m.zero_()
        m[: nzb.size] = torch.from_numpy(nzb.astype(np.uint8)).to(ctx.device)
    red.reduce(k, marked=marked)
red.flush()
torch.cuda.synchronize()
results["reducer_5"] = red.result(5).cpu().numpy().copy()
results["reducer_6"] = red.result(6).cpu().numpy().copy()
results["reducer_info"] = np.array([(i["n_union"], i["capacity"], i["mode"]) for i in red.info], np.int64)
tr.close()
np.savez(out / f"rank{rank}.npz", **results)
print("worker", rank, "done")

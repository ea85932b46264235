"""The path's one exchange step through the C-ABI (cpm_comm_*, cpm_allreduce_grid*: RCCL on the caller's stream), on one
MI355X: a real RCCL communicator of size 1 (ncclCommInitRank / ncclCommInitAll), the collectives enqueued on the stream,
the touched-brick reduce, and the sharding layer's RCCL transport.  (Two ranks on one device are refused by RCCL; the
N > 1 logic -- shards, double buffering, waits -- runs on 2 gloo ranks in tests/test_distributed_cpu.py over the same
classes with the torch transport.)"""
import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_world1_rccl_communicator_and_collectives(ctx, cpm):
    torch = ctx.torch
    uid = ctx.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = ctx.comm_create(uid, 0, 1)
    assert comm.rank == 0 and comm.size == 1
    g = torch.rand(128 ** 3, device=ctx.device)
    want = g.clone()
    ctx.allreduce_grid(comm, g)                       # ncclAllReduce, in place, on torch's current stream
    side = torch.cuda.Stream(device=ctx.device)
    with torch.cuda.stream(side):                     # and on another stream
        side.wait_stream(torch.cuda.current_stream(ctx.device))
        ctx.allreduce_grid(comm, g)
    torch.cuda.synchronize()
    assert torch.equal(g, want)
    out = torch.zeros_like(g)
    ctx.reduce_grid(comm, g, out, 0)
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    with pytest.raises(cpm.binding.CpmError):
        ctx.reduce_grid(comm, g, out, 3)              # root out of range
    comm.close()


def test_touched_brick_reduce(ctx, cpm):
    torch = ctx.torch
    B = cpm.binding
    comm = ctx.comm_create(ctx.comm_unique_id(), 0, 1)
    for dims, ch in (((64, 64, 64), 1), ((30, 18, 9), 4)):
        grid = B.default_grid_desc(dims, ch)
        cells = dims[0] * dims[1] * dims[2]
        bxn, byn, bzn = [(d + 3) // 4 for d in dims]
        nb = bxn * byn * bzn
        rng = np.random.default_rng(nb)
        mask_np = (rng.random(nb) < 0.05).astype(np.uint8)
        partial = torch.rand((cells, ch) if ch > 1 else (cells,), device=ctx.device)
        total = torch.full_like(partial, -1.0)
        mask = torch.from_numpy(mask_np).to(ctx.device)
        n_union = ctx.allreduce_grid_bricks(comm, partial, total, grid, mask)
        torch.cuda.synchronize()
        assert n_union == int(mask_np.sum())
        # exactly the voxels of the marked bricks were replaced by the (one-rank) sum
        z, y, x = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
        b = (x // 4) + bxn * ((y // 4) + byn * (z // 4))
        sel = mask_np[b.reshape(-1)].astype(bool)
        got, p = total.cpu().numpy().reshape(cells, -1), partial.cpu().numpy().reshape(cells, -1)
        assert np.array_equal(got[sel], p[sel]) and (got[~sel] == -1.0).all()
        # a mask beyond a quarter of the bricks: still correct for one rank (dense path needs > 1 rank)
        mask.fill_(1)
        n_union = ctx.allreduce_grid_bricks(comm, partial, total, grid, mask)
        torch.cuda.synchronize()
        assert n_union == nb and torch.equal(total, partial)
    comm.close()


def test_single_process_form(ctx, cpm):
    """cpm_comm_create_all / cpm_allreduce_grids: one host thread driving the devices (Inviwo's process model)."""
    torch = ctx.torch
    lib = ctx.lib
    ctxs = (C.c_void_p * 1)(ctx.h)
    comms = (C.c_void_p * 1)()
    assert lib.cpm_comm_create_all(ctxs, 1, comms) == 0
    assert lib.cpm_comm_size(comms[0]) == 1 and lib.cpm_comm_rank(comms[0]) == 0
    g = torch.rand(4096, device=ctx.device)
    want = g.clone()
    grids = (C.c_void_p * 1)(g.data_ptr())
    streams = (C.c_void_p * 1)(torch.cuda.current_stream(ctx.device).cuda_stream)
    assert lib.cpm_allreduce_grids(ctxs, comms, grids, g.numel(), streams, 1) == 0
    torch.cuda.synchronize()
    assert torch.equal(g, want)
    lib.cpm_comm_destroy(comms[0])


def test_rccl_transport_of_the_sharding_layer(ctx, cpm):
    torch = ctx.torch
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    tr = sh.RcclTransport(ctx, 0, 1)
    g = torch.rand(1 << 20, device=ctx.device)
    want = g * 3.0
    g.mul_(3.0)                                       # enqueued before the reduce: the side stream must wait for it
    handle = tr.start(g)
    tr.wait(handle)
    g2 = g.clone()                                    # enqueued after the wait: sees the reduced grid
    torch.cuda.synchronize()
    assert torch.equal(g2, want)
    red = sh.OverlappedGridReducer(g, tr)
    assert not red.active and red.acquire(0) is g
    tr.close()


def _bricks_of(dims):
    bxn, byn, bzn = [(d + 3) // 4 for d in dims]
    z, y, x = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
    return (x // 4) + bxn * ((y // 4) + byn * (z // 4)), bxn * byn * bzn


@pytest.mark.parametrize("dims,ch", [((128, 128, 128), 1), ((64, 32, 48), 4), ((30, 18, 9), 1), ((21, 7, 5), 4), ((256, 256, 48), 1)])
def test_sparse_grid_reduce_one_rank(ctx, cpm, dims, ch):
    """cpm_allreduce_grid_sparse with a real RCCL communicator of size 1: mask -> union -> list -> pack -> (sum) -> unpack, no
    host wait in the call; the union count arrives through the mailbox; in place the grid is unchanged bit for bit, a
    separate total is the grid on the union and zero elsewhere; a payload too small for the union falls back to the dense
    sum in cpm_sparse_reduce_complete; the automatic capacity follows the policy with the union of two calls before."""
    torch = ctx.torch
    B = cpm.binding
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    comm = ctx.comm_create(ctx.comm_unique_id(), 0, 1)
    gd = B.default_grid_desc(dims, ch)
    sr = ctx.sparse_reduce_create(comm, gd)
    cells = dims[0] * dims[1] * dims[2]
    b, nb = _bricks_of(dims)
    assert sr.n_bricks == nb
    rng = np.random.default_rng(nb + ch)
    lit_bricks = rng.random(nb) < 0.12
    vox = lit_bricks[b.reshape(-1)] & (rng.random(cells) < 0.4)
    g_np = np.zeros((cells, ch), np.float32)
    g_np[vox] = rng.random((int(vox.sum()), ch), dtype=np.float32) + 0.01
    g_np[vox, ch - 1] *= rng.random(int(vox.sum())) < 0.5   # zeros inside lit voxels too
    want_union = int(np.unique(b.reshape(-1)[(g_np != 0).any(axis=1)]).size)
    g = torch.from_numpy(g_np.reshape(-1)).to(ctx.device)
    keep = g.clone()
    # in place
    t1 = sr.start(g)
    i1 = sr.complete(t1)
    torch.cuda.synchronize()
    assert i1.n_union == want_union and i1.n_bricks == nb
    assert i1.capacity == sh.sparse_capacity(nb, -1)
    assert i1.mode == (1 if i1.capacity >= nb else 2 if want_union > i1.capacity else 0)
    assert torch.equal(g.view(torch.int32), keep.view(torch.int32))
    # a separate total: the grid on the union, zeros elsewhere (it held garbage)
    total = torch.full_like(g, 7.0)
    t2 = sr.start(g, total)
    i2 = sr.complete(t2)
    torch.cuda.synchronize()
    assert torch.equal(total.view(torch.int32), keep.view(torch.int32)) and i2.n_union == want_union
    # payload too small: overflow -> dense sum in complete()
    if want_union > 1:
        total.fill_(-3.0)
        t3 = sr.start(g, total, capacity=max(1, want_union // 2))
        i3 = sr.complete(t3)
        torch.cuda.synchronize()
        assert i3.mode == 2 and torch.equal(total.view(torch.int32), keep.view(torch.int32))
        assert i3.reduce_bytes == nb + i3.capacity * 256 * ch + cells * ch * 4 == nb + i3.capacity * 256 * ch + i3.dense_bytes
    else:
        sr.complete(sr.start(g, total))
    # the automatic capacity of call k comes from the union of call k - 2
    t4 = sr.start(g)
    i4 = sr.complete(t4)
    assert i4.capacity == sh.sparse_capacity(nb, want_union)
    if i4.mode == 0:
        assert i4.reduce_bytes == nb + i4.capacity * 256 * ch and i4.reduce_bytes < i4.dense_bytes
    # the caller's mask (delta path): exactly the marked bricks are replaced, the rest of a separate total is kept
    mask_np = (rng.random(nb) < 0.07).astype(np.uint8)
    mask = torch.from_numpy(mask_np).to(ctx.device)
    total.fill_(-1.0)
    i5 = sr.complete(sr.start(g, total, brick_mask=mask, capacity=nb // 2 if nb >= 4 else 0))
    torch.cuda.synchronize()
    if i5.mode == 0:
        sel = np.repeat(mask_np[b.reshape(-1)].astype(bool), ch)
        got = total.cpu().numpy()
        assert i5.n_union == int(mask_np.sum())
        assert np.array_equal(got[sel], g_np.reshape(-1)[sel]) and (got[~sel] == -1.0).all()
    torch.cuda.synchronize()
    assert torch.equal(g.view(torch.int32), keep.view(torch.int32))
    # the caller's NON-ZERO mask (cpm_gather_fast_marked's): as without a mask, a separate total is zero-filled elsewhere
    nzmask = np.zeros(nb, np.uint8)
    nzmask[np.unique(b.reshape(-1)[(g_np != 0).any(axis=1)])] = 1
    total.fill_(-5.0)
    i6 = sr.complete(sr.start(g, total, brick_mask=torch.from_numpy(nzmask).to(ctx.device), mask_is_nonzero=True, capacity=nb // 2 if nb >= 4 else 0))
    torch.cuda.synchronize()
    assert i6.n_union == want_union and torch.equal(total.view(torch.int32), keep.view(torch.int32))
    sr.close()
    comm.close()


def test_sparse_reduce_ticket_discipline(ctx, cpm):
    torch = ctx.torch
    B = cpm.binding
    comm = ctx.comm_create(ctx.comm_unique_id(), 0, 1)
    sr = ctx.sparse_reduce_create(comm, B.default_grid_desc((16, 16, 16), 1))
    g = torch.zeros(4096, device=ctx.device)
    g[5] = 1.0
    tickets = [sr.start(g) for _ in range(8)]
    with pytest.raises(B.CpmError):
        sr.start(g)                       # 8 issued and not completed
    for t in tickets:
        assert sr.complete(t).n_union == 1
    assert sr.complete(tickets[-1]).n_union == 1   # completing twice is harmless
    with pytest.raises(B.CpmError):
        sr.complete(tickets[-1] + 1)      # never issued
    t = sr.start(g)
    assert t == tickets[-1] + 1
    sr.complete(t)
    with pytest.raises(B.CpmError):
        sr.start(g, capacity=4096)        # beyond the number of bricks
    # an empty grid: union 0, nothing moves, the grid stays zero
    g.zero_()
    assert sr.complete(sr.start(g)).n_union == 0
    torch.cuda.synchronize()
    assert not g.any()
    sr.close()
    comm.close()


def test_overlapped_reducer_sparse_on_one_gpu(ctx, cpm):
    """The frame loop of bench.py with the sparse reduce forced on at one rank: every frame's grid comes back unchanged,
    and the figures of every ticket are recorded."""
    torch = ctx.torch
    B = cpm.binding
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    tr = sh.RcclTransport(ctx, 0, 1)
    gd = B.default_grid_desc((64, 64, 64), 1)
    first = torch.zeros(64 ** 3, device=ctx.device)
    red = sh.OverlappedGridReducer(first, tr, sparse=gd, force=True)
    assert red.active
    frames = []
    for k in range(6):
        out = red.acquire(k)
        out.zero_()
        out[: 64 * 64 * (2 + k)] = float(k + 1)
        frames.append(out.clone())
        if k % 2:      # odd frames hand over the non-zero marks a gather would have written
            m = red.marks_for(k)
            m.zero_()
            m[: 256 * ((2 + k + 3) // 4)] = 1
        red.reduce(k, marked=bool(k % 2))
    red.flush()
    torch.cuda.synchronize()
    assert torch.equal(red.result(5), frames[5]) and torch.equal(red.result(4), frames[4])
    assert len(red.info) == 6
    assert [i["n_union"] for i in red.info] == [256 * ((2 + k + 3) // 4) for k in range(6)]
    tr.close()


def test_bricklist_ticket_discipline_with_one_rank(ctx, cpm):
    """cpm_bricklist_reduce_* over a real RCCL communicator of ONE rank: nothing travels, every call still has to come in order -- open (no segment: the
    only rank is the root and gathers into its grid), exchange once, complete; four open tickets at most; the point-to-point calls refuse a peer that
    is not there."""
    B = cpm.binding
    torch = ctx.torch
    comm = ctx.comm_create(ctx.comm_unique_id(), 0, 1)
    gd = B.default_grid_desc((16, 12, 8), 1)
    br = ctx.bricklist_reduce_create(comm, gd, 0)
    g = torch.ones(16 * 12 * 8, dtype=torch.float32, device=ctx.device)
    t1, seg = br.open()
    assert t1 == 1 and not seg.segment and seg.room == 64 and seg.channels == 1
    br.pack_grid(t1, g)                        # (nothing to pack at the root)
    with pytest.raises(B.CpmError):
        br.complete(t1)                        # opened, never exchanged -- with one rank there is nothing to wait for, but the order is the order
    br.exchange(t1, g)
    with pytest.raises(B.CpmError):
        br.exchange(t1, g)                     # once
    i = br.complete(t1)
    assert (i.n_own, i.capacity, i.resent, i.sent_bytes, i.received_bytes, i.listed_bricks) == (0, 0, 0, 0, 0, 0) and i.n_bricks == 4 * 3 * 2
    assert torch.equal(g, torch.ones_like(g))
    with pytest.raises(B.CpmError):
        br.complete(99)
    with pytest.raises(B.CpmError):
        br.exchange(2, g)                      # not opened yet
    tickets = [br.open()[0] for _ in range(4)]
    with pytest.raises(B.CpmError):
        br.open()                              # four open and not completed
    for t in tickets:
        br.exchange(t, g)
        br.complete(t)
    t6 = br.start(g)                           # the one-call form still works beside the step-by-step one
    assert t6 == 6
    br.complete(t6)
    buf = torch.zeros(64, dtype=torch.uint8, device=ctx.device)
    with pytest.raises(B.CpmError):
        ctx.comm_send(comm, buf, 64, 0)        # a rank does not send to itself
    with pytest.raises(B.CpmError):
        ctx.comm_recv(comm, buf, 64, 1)        # nor receive from a rank that does not exist
    br.close()
    comm.close()

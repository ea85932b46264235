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

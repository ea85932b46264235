"""cpm_gather_fast_segment and cpm_bricklist_pack_grid on a real MI355X, one process (include/cpm/cpm.h, "the same exchange WITHOUT a
dense grid on the senders").

A segment is a public POD (cpm_bricklist_segment: buffer, capacity, room, ticket, the two control words, a mailbox word), so the test
builds one from plain device buffers -- no communicator -- and checks, through the C-ABI:
  * the segment the gather writes holds exactly the non-zero 4x4x4 bricks of the volume the dense launch (cpm_gather_fast) stores on the
    same records: every id once, count in the header and the mailbox word, the control words back at zero;
  * added into a zero grid (cpm_bricklist_segment_to_grid) it IS that volume, bit for bit -- narrow boxes (config 2's r), the 3-candidate
    loops, the workspace's wide 6 x 6 x 2 box (tiles with a halo + merge), 4 channels, ragged grids, clustered photons;
  * a second launch into the same segment gives the same set (slot order may differ: the root adds by brick id).
The multi-rank exchange over these segments: tests/test_fake_rccl_gpu.py."""
import ctypes as C

import numpy as np
import pytest

from test_fast_gpu import _n, _t, make_photons

pytestmark = pytest.mark.gpu


def _segment(ctx, B, nb, ch, capacity, ticket):
    torch = ctx.torch
    room = (nb + 63) & ~63
    slot = 16 + 256 * ch
    buf = torch.full((16 + room * slot,), 0xAB, dtype=torch.uint8, device=ctx.device)   # (stale bytes: nothing may depend on zeros)
    ctl = torch.zeros(2, dtype=torch.int32, device=ctx.device)
    mail = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    seg = B.BricklistSegment(buf.data_ptr(), capacity, room, ticket, ch, ctl.data_ptr(), mail.data_ptr())
    return seg, buf, ctl, mail


def _parse(buf, ch):
    raw = buf.cpu().numpy()
    count, capacity, ticket, magic = raw[:16].view(np.uint32)
    slot = 16 + 256 * ch
    body = raw[16:16 + int(count) * slot].reshape(int(count), slot)
    ids = body[:, :4].copy().view(np.uint32).reshape(-1)
    vals = body[:, 16:].copy().view(np.float32).reshape(int(count), 64 * ch)
    return int(count), int(capacity), int(ticket), int(magic), ids, vals


def _brick_ids(dims):
    bxn, byn = (dims[0] + 3) // 4, (dims[1] + 3) // 4
    z, y, x = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
    return ((x // 4) + bxn * ((y // 4) + byn * (z // 4))).reshape(-1)


CASES = [
    # dims, channels, radius (texture units), photons, cluster
    ((128, 128, 128), 1, 0.5 / 128, 200000, 0.3),      # config 2's half-voxel radius: 2 candidates per axis
    ((64, 64, 64), 1, 1.2 / 64, 60000, None),          # 3 candidates per axis
    ((40, 28, 20), 4, 0.9 / 40, 30000, 0.5),           # 4 channels, ragged, anisotropic candidates
    ((19, 13, 9), 1, 0.6 / 19, 5000, None),            # not multiples of 4: bricks hanging over the faces
    ((256, 256, 48), 1, 2.8 / 256, 150000, 0.4),       # the workspace's 6 x 6 x 2 box: tiles with a halo + merge
    ((64, 64, 24), 4, 2.2 / 64, 40000, None),          # wide box, 4 channels
]


@pytest.mark.parametrize("dims,ch,radius,n,cluster", CASES)
def test_gather_into_a_segment_is_the_dense_gather(ctx, cpm, dims, ch, radius, n, cluster):
    B = cpm.binding
    torch = ctx.torch
    rng = np.random.default_rng(dims[0] * 7 + ch)
    grid = B.default_grid_desc(dims, ch)
    assert ctx.gather_fast_supported_on(grid, radius)
    ph = make_photons(rng, n, dims, cluster=cluster)
    d_ph = _t(ctx, ph)
    table = torch.zeros(ctx.fast_table_entries(grid, n), dtype=torch.int32, device=ctx.device)
    cap = ctx.fast_record_capacity(grid, n, radius)
    assert cap > 0
    srt = torch.empty((cap, 4 if ch == 1 else 8), dtype=torch.float32, device=ctx.device)
    ctx.bin_fast(d_ph, n, grid, radius, table, srt, layout=B.CPM_PHOTONS_INTERLEAVED)
    cells = dims[0] * dims[1] * dims[2]
    dense = torch.full((cells * ch,), -7.0, dtype=torch.float32, device=ctx.device)
    scale = 0.37
    ctx.gather_fast(srt, table, n, grid, radius, scale, dense)
    nb = ((dims[0] + 3) // 4) * ((dims[1] + 3) // 4) * ((dims[2] + 3) // 4)
    want = _n(dense).reshape(-1, ch)
    lit = np.unique(_brick_ids(dims)[(want != 0).any(axis=1)])
    assert 0 < lit.size <= nb
    sets = []
    for launch in range(2):
        seg, buf, ctl, mail = _segment(ctx, B, nb, ch, capacity=64, ticket=41 + launch)   # (a capacity far below the count: the gather still writes everything)
        ctx.gather_fast_segment(srt, table, n, grid, radius, scale, seg)
        torch.cuda.synchronize()
        count, capacity, ticket, magic, ids, vals = _parse(buf, ch)
        assert (capacity, ticket, magic) == (64, 41 + launch, 0x62726b32)
        assert count == lit.size and np.array_equal(np.sort(ids), lit)                     # every non-zero brick, once
        assert _n(ctl).tolist() == [0, 0]                                                  # ready for the next launch
        assert int(_n(mail)[0]) == ((41 + launch) << 32) | count
        back = torch.zeros(cells * ch, dtype=torch.float32, device=ctx.device)
        ctx.bricklist_segment_to_grid(seg, grid, back)
        torch.cuda.synchronize()
        assert np.array_equal(_n(back).view(np.uint32), _n(dense).view(np.uint32))         # the dense launch's volume, bit for bit
        assert (vals != 0).any(axis=1).all()
        sets.append(dict(zip(ids.tolist(), map(bytes, vals))))
    assert sets[0] == sets[1]


@pytest.mark.parametrize("dims,ch", [((32, 32, 32), 1), ((20, 13, 9), 4), ((64, 64, 64), 1)])
@pytest.mark.parametrize("with_marks", [False, True])
def test_pack_grid_lists_the_nonzero_bricks(ctx, cpm, dims, ch, with_marks):
    """The sender's launch from a DENSE grid (bricklist_pack_grid_kernel: what cpm_reduce_grid_bricklists / cpm_bricklist_pack_grid run): a
    communicator of one rank has no segment, so the launch is reached through its test hook (cpm_profile.h) on a caller-made segment --
    with and without the gather's marks the segment's bricks are the grid's non-zero ones, values bit for bit."""
    B = cpm.binding
    torch = ctx.torch
    rng = np.random.default_rng(dims[0] + 31 * ch)
    cells = dims[0] * dims[1] * dims[2]
    g = np.zeros((cells, ch), np.float32)
    lit_vox = rng.random(cells) < 0.02
    g[lit_vox] = rng.random((int(lit_vox.sum()), ch), dtype=np.float32) + np.float32(0.1)
    if ch == 4:
        g[:, 3] = 0
    grid = B.default_grid_desc(dims, ch)
    nb = ((dims[0] + 3) // 4) * ((dims[1] + 3) // 4) * ((dims[2] + 3) // 4)
    lit = np.unique(_brick_ids(dims)[(g != 0).any(axis=1)])
    d_g = _t(ctx, g.reshape(-1))
    marks = None
    if with_marks:
        m = np.zeros(nb + 16, np.uint8)
        m[lit] = 1
        marks = _t(ctx, m)
    lib = B.load_library()
    seg, buf, ctl, mail = _segment(ctx, B, nb, ch, capacity=64, ticket=5)
    lib.cpm_debug_pack_grid_segment.restype = C.c_int
    lib.cpm_debug_pack_grid_segment.argtypes = [C.c_void_p, C.POINTER(B.BricklistSegment), C.POINTER(B.GridDesc), C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib.cpm_debug_pack_grid_segment(ctx.h, C.byref(seg), C.byref(grid), ctx._ptr(d_g), ctx._ptr(marks) if marks is not None else None, ctx._stream())
    assert rc == 0
    torch.cuda.synchronize()
    count, _, _, _, ids, vals = _parse(buf, ch)
    assert count == lit.size and np.array_equal(np.sort(ids), lit)
    back = torch.zeros(cells * ch, dtype=torch.float32, device=ctx.device)
    ctx.bricklist_segment_to_grid(seg, grid, back)
    torch.cuda.synchronize()
    assert np.array_equal(_n(back).view(np.uint32), g.reshape(-1).view(np.uint32))


def test_root_adds_all_segments_in_rank_order(ctx, cpm):
    """The root's two launches (brick -> slot tables, then the sum) over FOUR senders' segments at once, through their measurement hook:
    real photons under tile shards, so every sender lists nearly every lit brick and most bricks have five contributors with inexact
    values -- the result is numpy's ((((own + s0) + s1) + s2) + s3) bit for bit; a segment whose list outgrew its capacity adds nothing
    (that sender goes again at exact size: the multi-rank tests), stale table entries of an earlier pass are not believed."""
    import importlib
    B, S, P = cpm.binding, cpm.synthetic, cpm.pipeline
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    torch = ctx.torch
    lib = B.load_library()
    lib.cpm_debug_root_add_segments.restype = C.c_int
    lib.cpm_debug_root_add_segments.argtypes = [C.c_void_p, C.POINTER(B.BricklistSegment), C.c_int, C.POINTER(B.GridDesc), C.c_void_p, C.c_void_p, C.c_void_p]
    vol_np, tf = S.heterogeneous_volume(64), S.workspace_tf()
    world, n_total, dims = 5, 256 * 256, (64, 64, 64)
    nb = 16 ** 3
    dense, segs, keep = [], [], []
    for r in range(world):
        fr = P.PhotonFrame(ctx, vol_np, tf, (256, 256), dims, light_travel_direction=(0.3, 0.5, -1.0), photon_indices=sh.shard_tiles(n_total, r, world, tile=1024))
        fr.set_planar_records(True)
        fr.trace(); fr.bin_fast(); fr.gather_fast()
        dense.append(_n(fr.light_volume).copy())
        if r > 0:
            seg, buf, ctl, mail = _segment(ctx, B, nb, 1, capacity=(nb + 63) & ~63, ticket=9)
            fr.gather_fast_segment(seg)
            segs.append(seg); keep.append((buf, ctl, mail))
        grid_desc = fr.grid
        torch.cuda.synchronize()
        del fr
    lit = [np.unique(_brick_ids(dims)[d != 0]) for d in dense]
    shared = lit[0]
    for l in lit[1:]:
        shared = np.intersect1d(shared, l)
    assert shared.size > 200                                   # bricks every rank lights
    slot_of = torch.zeros(5 * nb, dtype=torch.int32, device=ctx.device)   # (4 tables + the who-lists-what words; zeros in a table name slot 0: believed only where that slot carries the brick)
    arr = (B.BricklistSegment * 4)(*segs)
    for attempt in range(2):                                   # (the second pass finds the first one's table entries: all still true)
        total = _t(ctx, dense[0])
        assert lib.cpm_debug_root_add_segments(ctx.h, arr, 4, C.byref(grid_desc), ctx._ptr(total), ctx._ptr(slot_of), ctx._stream()) == 0
        torch.cuda.synchronize()
        want = dense[0].copy()
        for r in range(1, world):
            want = want + dense[r]
        assert np.array_equal(_n(total).view(np.uint32), want.view(np.uint32))
        assert not _n(slot_of[4 * nb:]).any()                   # the pass cleared its who-lists-what words
    # sender 2's segment "arrives" with a capacity its list outgrew: it adds nothing, the others add as before -- and the table entries it
    # left in the passes above are not believed
    segs2 = list(segs)
    s2 = segs[1]
    segs2[1] = B.BricklistSegment(s2.segment, 64, s2.room, s2.ticket, s2.channels, s2.control, s2.mailbox)
    # (the header still says the capacity it was filled for: make it this one's, as a sender that was told 64 would have written)
    hdr = keep[1][0][:16].cpu().numpy().view(np.uint32).copy()
    assert hdr[0] > 64
    hdr[1] = 64
    keep[1][0][:16] = torch.from_numpy(hdr.view(np.uint8)).to(ctx.device)
    total = _t(ctx, dense[0])
    assert lib.cpm_debug_root_add_segments(ctx.h, (B.BricklistSegment * 4)(*segs2), 4, C.byref(grid_desc), ctx._ptr(total), ctx._ptr(slot_of), ctx._stream()) == 0
    torch.cuda.synchronize()
    want = dense[0].copy()
    for r in (1, 3, 4):
        want = want + dense[r]
    assert np.array_equal(_n(total).view(np.uint32), want.view(np.uint32))

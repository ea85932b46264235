"""The N > 1 path on CPU: world_size-2 gloo.  Each rank owns a photon shard, builds its own full-size
irradiance grid and the grids are summed with one all-reduce -- the structure bench.py runs on GPUs.
The per-rank compute is done by the oracle here (no GPU in this container); what is under test is
the sharding and the collective: shards tile the photon range, photon i does not depend on the
shard it lands in, and the reduced grid equals the unsharded one within summation-order tolerance."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent


def test_shard_ranges_tile(cpm):
    sr = cpm.sharding.shard_range if hasattr(cpm, "sharding") else None
    import importlib
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    for n, w in ((1 << 20, 8), (10, 3), (7, 8), (0, 2)):
        r = [sh.shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
        assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    with pytest.raises(ValueError):
        sh.shard_range(10, 2, 2)


def test_shard_tiles_partition(cpm):
    """Tile-interleaved shards (bench.py's default): rank r owns the 4096-sample tiles t = r (mod N); the shards partition
    the photon range, are ascending, and differ by at most one tile."""
    import importlib
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    for n, w, tile in ((1 << 20, 8, 4096), (15360, 2, 4096), (10, 3, 4), (7, 8, 4096), (0, 2, 4096), (8192, 2, 4096), (12289, 3, 4096)):
        shards = [sh.shard_tiles(n, k, w, tile) for k in range(w)]
        allidx = np.concatenate(shards) if shards else np.zeros(0, np.int64)
        assert np.array_equal(np.sort(allidx), np.arange(n))
        assert all(np.all(np.diff(s) > 0) for s in shards if s.size > 1)
        assert max(s.size for s in shards) - min(s.size for s in shards) <= tile
        for k, s in enumerate(shards):
            assert np.all((s // tile) % w == k)
    assert np.array_equal(sh.shard_tiles(1 << 20, 0, 1), np.arange(1 << 20))
    with pytest.raises(ValueError):
        sh.shard_tiles(10, 2, 2)


def _lattice(world, scaling):
    """weak: every rank brings its own 96 x 96 lattice rows (bench.py --scaling weak); strong: ONE fixed lattice whatever
    the number of ranks (BASELINE configs 4 and 5: a fixed photon count sharded over the GPUs)."""
    return (96, 96 * world) if scaling == "weak" else (96, 160)


def _worker(rank, world, port, out_dir, scaling, shards="range"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(REPO))
    sys.path.insert(0, str(REPO / "tests"))
    import importlib
    import torch
    import torch.distributed as dist
    import cpm_amd
    from oracle_binding import Oracle, OTraceParams
    sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, P = cpm_amd.synthetic, cpm_amd.pipeline
    o = Oracle()
    nx, ny = _lattice(world, scaling)
    n_total = nx * ny
    if shards == "tiles":   # (tiles of 1024 samples: the small test lattice then has several tiles per rank)
        gidx = sh.shard_tiles(n_total, rank, world, 1024)
    else:
        lo, hi = sh.shard_range(n_total, rank, world)
        gidx = np.arange(lo, hi, dtype=np.int64)
    lo, hi = 0, int(gidx.size)   # local extent
    vol_np, tf = S.heterogeneous_volume(32), S.workspace_tf()
    d = P._normalize((0.3, 0.5, -1.0))
    origin = np.array([0.5] * 3, np.float32) - np.float32(2) * d
    po, u, v = P.fit_plane_aligned_obb(S.UNIT_CUBE_VERTICES, origin, d)
    area = float(np.float32(np.linalg.norm(u)) * np.float32(np.linalg.norm(v)))
    s = o.uniform_samples_2d(nx, ny)[gidx].copy()
    ls = o.directional_light_samples(s, (1, 1, 1), d, po, u, v, area)
    isect = o.light_sample_box_intersection(ls, S.UNIT_CUBE_AABB)
    st = np.zeros((n_total, 2), np.uint32)
    st[:, 0] = o.glibc_rand_sequence(0, n_total)
    o.seed_streams(st, 1 << 40)
    st = st[gidx].copy()
    p = OTraceParams()
    p.step_size = 1 / 32
    p.n_light_samples = hi - lo
    p.max_interactions = 1
    p.total_photons = hi - lo
    ph = np.zeros((hi - lo, 8), np.float32)
    o.trace(o.volume(vol_np), tf, S.UNIT_CUBE_AABB, p, ls, isect, st, ph)
    og = o.grid((16, 16, 16), 1)
    radius = S.photon_radius_texture((32, 32, 32), 1.0)
    scale = o.relative_irradiance_scale(radius, n_total)  # normalised by the photons of ALL ranks
    _, cs, srt = o.bin(ph, hi - lo, og)
    grid = np.zeros(16 ** 3, np.float32)
    o.gather(srt, cs, hi - lo, og, radius, scale, grid)
    t = torch.from_numpy(grid)
    sh.allreduce_light_volume(t)                           # the one collective of the path
    np.save(os.path.join(out_dir, f"photons_{rank}.npy"), ph)
    np.save(os.path.join(out_dir, f"index_{rank}.npy"), gidx)
    if rank == 0:
        np.save(os.path.join(out_dir, "grid.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scaling,shards", [("weak", "range"), ("strong", "range"), ("weak", "tiles"), ("strong", "tiles")])
def test_two_rank_photon_sharding_and_grid_allreduce(tmp_path, oracle, cpm, scaling, shards):
    import socket
    import torch.multiprocessing as mp
    from oracle_binding import OTraceParams
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    mp.spawn(_worker, args=(world, port, str(tmp_path), scaling, shards), nprocs=world, join=True)
    # unsharded reference
    S, P = cpm.synthetic, cpm.pipeline
    nx, ny = _lattice(world, scaling)
    n = nx * ny
    vol_np, tf = S.heterogeneous_volume(32), S.workspace_tf()
    d = P._normalize((0.3, 0.5, -1.0))
    origin = np.array([0.5] * 3, np.float32) - np.float32(2) * d
    po, u, v = P.fit_plane_aligned_obb(S.UNIT_CUBE_VERTICES, origin, d)
    area = float(np.float32(np.linalg.norm(u)) * np.float32(np.linalg.norm(v)))
    smp = oracle.uniform_samples_2d(nx, ny)
    ls = oracle.directional_light_samples(smp, (1, 1, 1), d, po, u, v, area)
    isect = oracle.light_sample_box_intersection(ls, S.UNIT_CUBE_AABB)
    st = np.zeros((n, 2), np.uint32)
    st[:, 0] = oracle.glibc_rand_sequence(0, n)
    oracle.seed_streams(st, 1 << 40)
    p = OTraceParams()
    p.step_size = 1 / 32
    p.n_light_samples = n
    p.max_interactions = 1
    p.total_photons = n
    ph = np.zeros((n, 8), np.float32)
    oracle.trace(oracle.volume(vol_np), tf, S.UNIT_CUBE_AABB, p, ls, isect, st, ph)
    sharded = np.zeros_like(ph)
    seen = np.zeros(n, np.int32)
    for r in range(world):   # local photon j of rank r is global photon index_r[j]
        gi = np.load(tmp_path / f"index_{r}.npy")
        sharded[gi] = np.load(tmp_path / f"photons_{r}.npy")
        seen[gi] += 1
    assert np.all(seen == 1)
    assert np.array_equal(sharded.view(np.uint32), ph.view(np.uint32))  # photon i is shard-independent
    og = oracle.grid((16, 16, 16), 1)
    radius = S.photon_radius_texture((32, 32, 32), 1.0)
    scale = oracle.relative_irradiance_scale(radius, n)
    _, cs, srt = oracle.bin(ph, n, og)
    want = np.zeros(16 ** 3, np.float32)
    oracle.gather(srt, cs, n, og, radius, scale, want)
    got = np.load(tmp_path / "grid.npy")
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6 * float(want.max()))
    assert want.sum() > 0


def _overlap_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(REPO))
    import importlib
    import torch
    import torch.distributed as dist
    import cpm_amd
    sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first = torch.zeros(4096)
    red = sh.OverlappedGridReducer(first)
    assert red.active and red.buffers[0] is first
    results = []
    K = 7
    for k in range(K):
        # "trace + bin" of frame k would run here, overlapping the reduce of frame k - 1
        out = red.acquire(k)                       # waits for the reduce of frame k - 2 (same buffer)
        if k >= 2:
            results.append((k - 2, out.clone()))   # ... so that frame's reduced grid is complete here
        out.copy_(torch.full((4096,), float((rank + 1) * (k + 1))))   # the "gather" of frame k
        red.reduce(k)
    red.flush()
    results.append((K - 2, red.result(K - 2).clone()))
    results.append((K - 1, red.result(K - 1).clone()))
    want_scale = sum(r + 1 for r in range(world))
    ok = all(bool((g == want_scale * (k + 1)).all()) for k, g in results) and len(results) == K
    with open(os.path.join(out_dir, f"ok_{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_grid_reduce_double_buffering(tmp_path, cpm):
    """bench.py's collective: asynchronous all-reduce of frame k's grid while frame k + 1 is computed, two
    buffers, the wait placed where a buffer is reused.  Every frame's reduced grid must be the rank sum."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_overlap_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok_0").read_text() == "1" and (tmp_path / "ok_1").read_text() == "1"


def test_overlapped_grid_reduce_without_process_group(cpm):
    import importlib
    import torch
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    red = sh.OverlappedGridReducer(torch.zeros(8))
    assert not red.active
    for k in range(3):
        red.acquire(k).fill_(k + 1.0)
        red.reduce(k)
    red.flush()
    assert float(red.result(2)[0]) == 3.0 and float(red.result(1)[0]) == 2.0


def _root_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(REPO))
    import importlib
    import torch
    import torch.distributed as dist
    import cpm_amd
    sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    red = sh.OverlappedGridReducer(torch.zeros(1000), sh.TorchTransport(root=0))   # bench.py --collective reduce
    ok = True
    for k in range(5):
        out = red.acquire(k)
        out.copy_(torch.full((1000,), float((rank + 1) * (k + 1))))
        red.reduce(k)
    red.flush()
    for k in (3, 4):
        got = red.result(k)
        want = float(sum(r + 1 for r in range(world)) * (k + 1)) if rank == 0 else None
        ok = ok and (bool((got == want).all()) if rank == 0 else True)   # only the root's grid is defined afterwards
    with open(os.path.join(out_dir, f"ok_{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_to_the_display_rank(tmp_path, cpm):
    """The same double-buffered reducer with a root: only rank 0 (the GPU that renders) receives the summed grid."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_root_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok_0").read_text() == "1" and (tmp_path / "ok_1").read_text() == "1"


def _sparse_worker(rank, world, port, out_dir, dims, channels, root):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(REPO))
    import importlib
    import torch
    import torch.distributed as dist
    import cpm_amd
    sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dx, dy, dz = dims
    n = dx * dy * dz * channels
    nb = ((dx + 3) // 4) * ((dy + 3) // 4) * ((dz + 3) // 4)
    gen = torch.Generator().manual_seed(1234 + rank)

    def partial_grid(k):
        """A rank's light volume of frame k: a lit slab whose depth grows with k (so the union outgrows a capacity sized two
        frames earlier), every rank lighting slightly different voxels of it; exactly representable values (sums are exact)."""
        g = torch.zeros(dz, dy, dx, channels)
        depth = min(dz, 2 + 7 * k if k < 5 else 4)
        lit = torch.rand(depth, dy, dx, generator=gen) < 0.3
        vals = torch.randint(1, 1000, (depth, dy, dx, channels), generator=gen).float() / 8.0
        g[:depth] = vals * lit[..., None]
        return g.reshape(-1)

    red = sh.OverlappedGridReducer(torch.zeros(n), sh.TorchTransport(root=root), sparse=(dims, channels))
    K = 8
    mine = []
    for k in range(K):
        out = red.acquire(k)
        mine.append(partial_grid(k))
        out.copy_(mine[-1])
        red.reduce(k)
        # dense reference of the same frame, computed with the plain collective
        want = mine[-1].clone()
        dist.all_reduce(want)
        if root is None or rank == root:
            got = red.result(k)
            assert torch.equal(got, want), f"frame {k}: sparse sum != dense sum"
    red.flush()
    infos = red.info
    assert len(infos) == K
    # the capacity of frame k is the policy applied to the union of frame k - 2; the union is the same on every rank
    for k, i in enumerate(infos):
        assert i["n_bricks"] == nb
        assert i["capacity"] == sh.sparse_capacity(nb, infos[k - 2]["n_union"] if k >= 2 else -1)
        assert i["mode"] == (1 if i["capacity"] >= nb else 2 if i["n_union"] > i["capacity"] else 0)
    unions = torch.tensor([i["n_union"] for i in infos])
    gathered = [torch.zeros_like(unions) for _ in range(world)]
    dist.all_gather(gathered, unions)
    assert all(torch.equal(g, unions) for g in gathered)
    with open(os.path.join(out_dir, f"modes_{rank}"), "w") as f:
        f.write(",".join(str(i["mode"]) for i in infos) + ";" + ",".join(str(i["reduce_bytes"]) for i in infos) + ";" + str(infos[0]["dense_bytes"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dims,channels,root", [((32, 32, 32), 1, None), ((20, 13, 9), 1, None), ((16, 16, 24), 4, None), ((32, 32, 32), 1, 0)])
def test_sparse_grid_reduce_equals_dense_sum(tmp_path, cpm, dims, channels, root):
    """cpm_allreduce_grid_sparse's steps over gloo (TorchTransport carries them out with torch ops): the sum over the union
    of the ranks' non-zero 4x4x4 bricks equals the dense sum bit for bit -- on the union and (zeros) elsewhere --, ragged
    grids and 4 channels included; the payload is sized by the policy from the union two frames before; a union that
    outgrows it falls back to the dense sum on every rank alike; steady frames move a fraction of the dense bytes."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sparse_worker, args=(2, port, str(tmp_path), dims, channels, root), nprocs=2, join=True)
    m0, m1 = (tmp_path / "modes_0").read_text(), (tmp_path / "modes_1").read_text()
    assert m0 == m1
    modes, moved, dense = m0.split(";")
    modes, moved, dense = [int(x) for x in modes.split(",")], [int(x) for x in moved.split(",")], int(dense)
    if dims == (32, 32, 32):
        assert 2 in modes[:5]      # the growing slab overflowed a payload sized for an earlier frame
        assert modes[-1] == 0 and moved[-1] * 2 < dense  # the steady thin slab: sparse, a fraction of the dense bytes


def _lists_worker(rank, world, port, out_dir, dims, channels, root, slabs, inexact=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(REPO))
    import importlib
    import torch
    import torch.distributed as dist
    import cpm_amd
    sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dx, dy, dz = dims
    n = dx * dy * dz * channels
    nb = ((dx + 3) // 4) * ((dy + 3) // 4) * ((dz + 3) // 4)
    gen = torch.Generator().manual_seed(99 + rank)

    def partial_grid(k):
        """A rank's light volume of frame k.  slabs: the rank lights ITS slab of y rows (a contiguous photon range: a slab of the light
        plane) and a border row of the next rank's -- brick sets nearly disjoint; else every rank lights the same region (tile shards).
        The lit depth grows with k and comes back, so a list outgrows the segment sized two frames earlier.  Values are multiples of
        1/8 below 2^10: sums are exact in any order."""
        g = torch.zeros(dz, dy, dx, channels)
        depth = min(dz, 2 + 10 * k if k < 4 else 4)
        lit = torch.rand(depth, dy, dx, generator=gen) < 0.35
        if slabs:
            rows = torch.zeros(dy, dtype=torch.bool)
            lo, hi = rank * dy // world, (rank + 1) * dy // world
            rows[lo:min(dy, hi + 1)] = True
            lit &= rows[None, :, None]
        vals = torch.randint(1, 1000, (depth, dy, dx, channels), generator=gen).float() / 8.0
        if inexact:   # values whose sums DO depend on the order: the root's sum is then held to the protocol's order, not to any all-reduce's
            vals = torch.rand(depth, dy, dx, channels, generator=gen) + 0.01
        g[:depth] = vals * lit[..., None]
        return g.reshape(-1)

    red = sh.OverlappedGridReducer(torch.zeros(n), sh.TorchTransport(), lists=(dims, channels), root=root)
    K = 8
    kept = []
    for k in range(K):
        out = red.acquire(k)
        mine = partial_grid(k)
        out.copy_(mine)
        red.reduce(k)
        got = red.result(k)
        if inexact:
            kept.append((mine, got.clone()))
        else:
            want = mine.clone()
            dist.all_reduce(want)     # the dense sum of the same frame (exact values: any order gives these bits)
            if rank == root:
                assert torch.equal(got, want), f"frame {k}: sum of the lists != dense sum"
        if rank != root:
            assert torch.equal(got, mine), f"frame {k}: a sender's grid was written"
    red.flush()
    infos = red.info
    assert len(infos) == K
    if inexact:
        # the library's order: the root's own volume, then the senders in rank order, a sender whose list had outgrown its segment after the others
        for k in range(K):
            parts = [torch.zeros(n) for _ in range(world)] if rank == root else None
            dist.gather(kept[k][0], parts, dst=root)
            late = torch.tensor([1.0 if (rank != root and infos[k]["resent"]) else 0.0])
            lates = [torch.zeros(1) for _ in range(world)]
            dist.all_gather(lates, late)
            if rank == root:
                order = [r for r in range(world) if r != root and not lates[r].item()] + [r for r in range(world) if r != root and lates[r].item()]
                want = parts[root].clone()
                for r in order:
                    want = want + parts[r]
                assert torch.equal(kept[k][1], want), f"frame {k}: not the sum in rank order (late: {[r for r in order if lates[r].item()]})"
    for k, i in enumerate(infos):
        assert i["n_bricks"] == nb
        if rank != root:   # a sender's capacity is the policy applied to ITS count of two frames before; a list that outgrew it went again
            assert i["capacity"] == sh.bricklist_capacity(nb, infos[k - 2]["n_own"] if k >= 2 else -1)
            assert i["resent"] == (1 if i["n_own"] > i["capacity"] else 0)
            assert i["sent_bytes"] >= sh.bricklist_segment_bytes(i["capacity"], channels) and i["received_bytes"] == 0
    own = torch.tensor([[i["n_own"], i["resent"], i["sent_bytes"], i["received_bytes"], i["listed_bricks"]] for i in infos])
    everyone = [torch.zeros_like(own) for _ in range(world)]
    dist.all_gather(everyone, own)
    if rank == root:
        for k in range(K):   # the root received what the others sent, and listed what they listed
            assert int(own[k, 3]) == sum(int(everyone[r][k, 2]) for r in range(world) if r != root)
            assert int(own[k, 4]) == sum(int(everyone[r][k, 0]) for r in range(world) if r != root)
            assert int(own[k, 1]) == sum(int(everyone[r][k, 1]) for r in range(world) if r != root)
        with open(os.path.join(out_dir, "root"), "w") as f:
            f.write(";".join(",".join(str(int(v)) for v in own[:, c]) for c in range(5)) + ";" + str(infos[0]["dense_bytes"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dims,channels,root,slabs,inexact", [(2, (32, 32, 32), 1, 0, True, False), (2, (20, 13, 9), 4, 1, False, False),
                                                                   (4, (32, 32, 32), 1, 0, True, False), (4, (16, 24, 16), 4, 2, True, True),
                                                                   (8, (32, 64, 16), 1, 0, True, False),
                                                                   (8, (32, 32, 32), 1, 3, False, True)])   # 8 ranks lighting the SAME bricks, inexact values, overflows
def test_brick_lists_to_the_root_equal_the_dense_sum(tmp_path, cpm, world, dims, channels, root, slabs, inexact):
    """cpm_reduce_grid_bricklists's protocol over gloo (TorchTransport carries it out with send / recv; slots in a shuffled order, as the
    library's counter hands them out), 2, 4 and 8 ranks: the root's grid becomes the dense sum bit for bit -- with INEXACT values (the 4-rank
    4-channel case and the 8-rank case whose ranks all light the same bricks) the sum in the protocol's order: the root's own, the senders
    in rank order, a sender whose list had outgrown its segment last --, the senders' grids are left alone; a segment is sized from the sender's count two frames before
    (sender and root derive the same number on their own), a list that outgrew it goes again at exact size between that pair alone; with
    slab shards the root receives a fraction of what the dense reduce moves."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_lists_worker, args=(world, port, str(tmp_path), dims, channels, root, slabs, inexact), nprocs=world, join=True)
    cols = (tmp_path / "root").read_text().split(";")
    resent, received, dense = [int(x) for x in cols[1].split(",")], [int(x) for x in cols[3].split(",")], int(cols[5])
    nb = ((dims[0] + 3) // 4) * ((dims[1] + 3) // 4) * ((dims[2] + 3) // 4)
    if nb >= 512 and (world <= 4 or not slabs):   # (a grid of a few dozen bricks, or an eighth of one, fits its first segment whatever happens)
        assert sum(resent[:5]) >= 1      # the growing slab outgrew a segment sized for an earlier frame
    assert resent[-1] == 0               # the steady thin slab fits
    if slabs and nb >= 512:            # (the + 64 bricks of head-room dominate a grid of a hundred bricks: the byte model sends that one densely)
        assert received[-1] * 2 < dense * (world - 1)   # ... and is a fraction of N - 1 dense grids

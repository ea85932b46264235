"""The C-ABI's multi-rank path with TWO ranks, on the one GPU of the test box: libcpm_hip binds the RCCL test double
(tests/fake_rccl: the eight RCCL entry points over shared memory; CPM_RCCL_LIBRARY) instead of librccl, which refuses two ranks on
one device.  What runs is the product's code -- cpm_comm_create with rank / size, cpm_allreduce_grid / cpm_reduce_grid,
cpm_allreduce_grid_sparse (union of two DIFFERENT masks, slot tables, the capacity policy on both ranks, overflow -> dense on both,
root reduce, the delta path), sharding.RcclTransport + OverlappedGridReducer -- and the sums are compared with numpy's.  Not a
substitute for a run over xGMI: RCCL itself has still never executed with more than one rank."""
import importlib
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "tests" / "fake_rccl"))


def _partial(dims, ch, k, who):
    dx, dy, dz = dims
    g = np.zeros((dz, dy, dx, ch), np.float32)
    rng = np.random.default_rng(1000 * k + 17 * who + dx)
    depth = min(dz, (2 + 7 * k) if k < 5 else 4)
    lit = rng.random((depth, dy, dx)) < 0.3
    vals = rng.integers(1, 1000, (depth, dy, dx, ch)).astype(np.float32) / np.float32(8.0)
    g[:depth] = vals * lit[..., None]
    return g.reshape(-1)


def _bricks(dims):
    bxn, byn, bzn = [(d + 3) // 4 for d in dims]
    z, y, x = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
    return ((x // 4) + bxn * ((y // 4) + byn * (z // 4))).reshape(-1), bxn * byn * bzn


@pytest.fixture(scope="module")
def two_ranks(tmp_path_factory, cpm):
    import build as fake_build
    lib = fake_build.build()
    out = tmp_path_factory.mktemp("fake_rccl")
    env = dict(os.environ, CPM_RCCL_LIBRARY=str(lib))
    procs = [subprocess.Popen([sys.executable, str(REPO / "tests" / "fake_rccl" / "worker.py"), str(r), "2", str(out)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("the two-rank workers did not finish in 300 s")
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    return [np.load(out / f"rank{r}.npz") for r in range(2)]


def test_dense_collectives_with_two_ranks(two_ranks):
    r0, r1 = two_ranks
    want = _partial((16, 16, 16), 1, 0, 0) + _partial((16, 16, 16), 1, 0, 1)
    assert np.array_equal(r0["dense_allreduce"], want) and np.array_equal(r1["dense_allreduce"], want)
    want = _partial((16, 16, 16), 1, 1, 0) + _partial((16, 16, 16), 1, 1, 1)
    assert np.array_equal(r0["dense_reduce_root0"], want) and (r1["dense_reduce_root0"] == -1.0).all()


@pytest.mark.parametrize("dims,ch", [((32, 32, 32), 1), ((20, 13, 9), 4)])
def test_sparse_reduce_with_two_ranks(two_ranks, cpm, dims, ch):
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    r0, r1 = two_ranks
    b, nb = _bricks(dims)
    key = f"sparse_{dims[0]}_{ch}"
    i0, i1 = r0[key + "_info"], r1[key + "_info"]
    assert np.array_equal(i0, i1)                                   # union, capacity, mode and bytes: the same numbers on both ranks
    for k in range(8):
        a, c = _partial(dims, ch, k, 0), _partial(dims, ch, k, 1)
        want = a + c
        assert np.array_equal(r0[f"{key}_{k}"], want), k            # in place (even k) and separate total (odd k): the dense sum, bit for bit
        assert np.array_equal(r1[f"{key}_{k}"], want), k
        union = np.unique(b[np.repeat(((a != 0) | (c != 0)).reshape(-1, ch).any(axis=1), 1)]).size
        assert i0[k, 0] == union
        assert i0[k, 1] == sh.sparse_capacity(nb, int(i0[k - 2, 0]) if k >= 2 else -1)
        assert i0[k, 2] == (1 if i0[k, 1] >= nb else 2 if union > i0[k, 1] else 0)
    if dims == (32, 32, 32):
        modes = list(i0[:, 2])
        assert 2 in modes and 1 in modes and modes[-1] == 0         # overflow -> dense, dense by policy, and back to sparse
    # root reduce: rank 1 has the sum, rank 0 its own partial
    a, c = _partial(dims, ch, 2, 0), _partial(dims, ch, 2, 1)
    assert np.array_equal(r1[key + "_root1"], a + c) and np.array_equal(r0[key + "_root1"], a)
    # delta path: the union of the two ranks' touched bricks carries the sum, the rest of the separate total is untouched
    m = (r0[key + "_delta_mask"] | r1[key + "_delta_mask"]).astype(bool)
    sel = np.repeat(m[b], ch)
    want = _partial(dims, ch, 3, 0) + _partial(dims, ch, 3, 1)
    for r in (r0, r1):
        got = r[key + "_delta"]
        assert r[key + "_delta_union"][0] == m.sum()
        if r[key + "_delta_union"][1] == 0:
            assert np.array_equal(got[sel], want[sel]) and (got[~sel] == -2.0).all()


@pytest.mark.parametrize("dims,ch", [((32, 32, 32), 1), ((20, 13, 9), 4)])
def test_one_rank_rebuilds_while_the_other_updates(two_ranks, dims, ch):
    """The frame in which the shards take different paths (a rank-local decision: ADVICE r04): the rebuilding rank hands in every brick
    it had lit or lights now, the updating rank its touched bricks, both as TOUCHED masks -- the standing sum becomes the sum of the
    two CURRENT partial volumes everywhere, on both ranks (what PhotonToLightVolumeProcessorCL::reduceOverShards now does)."""
    r0, r1 = two_ranks
    key = f"sparse_{dims[0]}_{ch}"
    want = r0[key + "_mixed_new"] + r1[key + "_mixed_new"]
    assert np.array_equal(r0[key + "_mixed"], want) and np.array_equal(r1[key + "_mixed"], want)
    assert not np.array_equal(r0[key + "_mixed_new"], _partial(dims, ch, 4, 0))   # rank 0 really rebuilt ...
    assert not np.array_equal(r1[key + "_mixed_new"], _partial(dims, ch, 4, 1))   # ... and rank 1 really changed something


def test_overlapped_reducer_over_two_ranks(two_ranks):
    """bench.py's frame loop: every frame's buffer holds the two ranks' sum once its reduce has been waited for, with and without the
    gather's marks; the figures of every ticket agree between the ranks."""
    r0, r1 = two_ranks
    dims = (32, 32, 32)
    for k in range(7):
        want = _partial(dims, 1, k, 0) + _partial(dims, 1, k, 1)
        assert np.array_equal(r0[f"reducer_{k}"], want), k
        assert np.array_equal(r1[f"reducer_{k}"], want), k
    assert np.array_equal(r0["reducer_info"], r1["reducer_info"]) and r0["reducer_info"].shape == (7, 3)


# ---- the brick-list exchange (cpm_reduce_grid_bricklists): 2 and 4 ranks on the one GPU -------------------------------------------------

def _slab_partial(dims, ch, k, who, n_ranks):
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


_LISTS_RUNS = {}


def _run_lists(tmp_path_factory, world, root):
    """The workers' results for (world, root): started once per session, shared by the tests that read them."""
    if (world, root) not in _LISTS_RUNS:
        _LISTS_RUNS[(world, root)] = _start_lists(tmp_path_factory, world, root)
    return _LISTS_RUNS[(world, root)]


def _start_lists(tmp_path_factory, world, root):
    import build as fake_build
    lib = fake_build.build()
    out = tmp_path_factory.mktemp(f"fake_rccl_lists_{world}")
    env = dict(os.environ, CPM_RCCL_LIBRARY=str(lib))
    procs = [subprocess.Popen([sys.executable, str(REPO / "tests" / "fake_rccl" / "worker_lists.py"), str(r), str(world), str(out), str(root)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail(f"the {world}-rank brick-list workers did not finish in 420 s")
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    return [np.load(out / f"rank{r}.npz") for r in range(world)]


@pytest.mark.parametrize("world,root", [(2, 0), (4, 1)])
def test_brick_lists_to_the_root(tmp_path_factory, cpm, world, root):
    """Every rank but the root packs ITS non-zero 4x4x4 bricks and sends that one segment to the root, which adds the segments in rank
    order: the root's grid is the dense sum bit for bit (numpy's), the senders' grids are untouched; a segment is sized from the sender's
    count two tickets before -- sender and root each derive it on their own --, a list that outgrew it goes again at exact size between
    that pair alone; the root's byte counts are the senders'; slab shards move a fraction of the dense grids.  At most 4 + this process
    use the GPU at once (the box allows 6)."""
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    ranks = _run_lists(tmp_path_factory, world, root)
    for dims, ch in (((32, 32, 32), 1), ((20, 13, 9), 4)):
        key = f"lists_{dims[0]}_{ch}"
        nb = _bricks(dims)[1]
        infos = [r[key + "_info"] for r in ranks]
        for k in range(8):
            parts = [_slab_partial(dims, ch, k, w, world) for w in range(world)]
            want = parts[root].copy()
            for w in range(world):      # the root adds in rank order (exact values: any order gives these bits, but this IS the order)
                if w != root:
                    want = want + parts[w]
            assert np.array_equal(ranks[root][f"{key}_{k}"], want), (dims, k)
            for w in range(world):
                if w != root:
                    assert np.array_equal(ranks[w][f"{key}_{k}"], parts[w]), (dims, k, w)
                    n_own, cap, resent, sent = (int(v) for v in infos[w][k, :4])
                    b, _ = _bricks(dims)
                    assert n_own == np.unique(b[(parts[w] != 0).reshape(-1, ch).any(axis=1)]).size
                    assert cap == sh.bricklist_capacity(nb, int(infos[w][k - 2, 0]) if k >= 2 else -1)
                    assert resent == (1 if n_own > cap else 0)
                    exact = (n_own + 63) & ~63
                    assert sent == sh.bricklist_segment_bytes(cap, ch) + (sh.bricklist_segment_bytes(exact, ch) if resent else 0)
            assert int(infos[root][k, 4]) == sum(int(infos[w][k, 3]) for w in range(world) if w != root)   # received == sent
            assert int(infos[root][k, 5]) == sum(int(infos[w][k, 0]) for w in range(world) if w != root)   # listed == the lists
            assert int(infos[root][k, 2]) == sum(int(infos[w][k, 2]) for w in range(world) if w != root)   # repeated exchanges
        if dims == (32, 32, 32):
            assert sum(int(infos[w][:, 2].sum()) for w in range(world) if w != root) >= 1   # a list outgrew its segment at least once ...
            assert int(infos[root][-1, 2]) == 0                                                # ... and the steady frame fits
            assert int(infos[root][-1, 4]) * 2 < int(infos[root][-1, 7]) * (world - 1)         # a fraction of N - 1 dense grids
    # the frame loop (two tickets in flight, reduce stream beside the frame's)
    for k in range(7):
        parts = [_slab_partial((32, 32, 32), 1, k, w, world) for w in range(world)]
        want = parts[root].copy()
        for w in range(world):
            if w != root:
                want = want + parts[w]
        assert np.array_equal(ranks[root][f"reducer_{k}"], want), k
        for w in range(world):
            if w != root:
                assert np.array_equal(ranks[w][f"reducer_{k}"], parts[w]), (k, w)


@pytest.mark.parametrize("world,root", [(2, 0), (4, 1)])
def test_senders_gather_straight_into_their_segments(tmp_path_factory, cpm, world, root):
    """Round 6: a rank that is not the display GPU has no dense light volume -- cpm_gather_fast_segment writes the non-zero 4x4x4 bricks of
    its gather into the ticket's segment, ONE send carries it, the root adds all received segments with two launches.  The root's volume is
    numpy's sum of the ranks' dense volumes (the same records gathered densely on every rank) in the library's order, bit for bit, from
    inexact values: the root's own, then the senders in rank order -- under tile shards EVERY sender lists every brick --, a sender whose list
    had outgrown its segment (the first tickets: capacity = a quarter of the bricks) after the others.  Same workers as above."""
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    ranks = _run_lists(tmp_path_factory, world, root)
    nb = 8 * 8 * 8
    overflowed = 0
    for kind in ("tiles", "range"):
        infos = [r[f"seg_{kind}_info"] for r in ranks]
        for k in range(5):
            parts = [ranks[w][f"seg_{kind}_{k}_dense"] for w in range(world)]
            late = [w for w in range(world) if w != root and int(infos[w][k, 2])]
            want = parts[root].copy()
            for w in [w for w in range(world) if w != root and w not in late] + late:
                want = want + parts[w]
            assert np.array_equal(ranks[root][f"seg_{kind}_{k}"].view(np.uint32), want.view(np.uint32)), (kind, k)
            b, _ = _bricks((32, 32, 32))
            for w in range(world):
                if w == root:
                    continue
                n_own, cap, resent, sent = (int(v) for v in infos[w][k, :4])
                assert n_own == np.unique(b[parts[w] != 0]).size
                assert cap == sh.bricklist_capacity(nb, int(infos[w][k - 2, 0]) if k >= 2 else -1)
                assert resent == (1 if n_own > cap else 0)
                assert sent == sh.bricklist_segment_bytes(cap, 1) + (sh.bricklist_segment_bytes((n_own + 63) & ~63, 1) if resent else 0)
                overflowed += resent
            assert int(infos[root][k, 4]) == sum(int(infos[w][k, 3]) for w in range(world) if w != root)
            assert int(infos[root][k, 5]) == sum(int(infos[w][k, 0]) for w in range(world) if w != root)
        assert int(infos[root][-1, 2]) == 0                       # the steady frame fits
    assert overflowed >= 1                                        # ... and some early list did not
    if world > 2:   # bricks with three and more contributors were really there (else rank order would not have been exercised)
        lit = [np.unique(_bricks((32, 32, 32))[0][ranks[w]["seg_tiles_4_dense"] != 0]) for w in range(world)]
        assert np.intersect1d(np.intersect1d(lit[0], lit[1]), lit[2]).size > 50
    # the frame loop: every frame's volume at the root is the same sum (the photons do not change from frame to frame)
    parts = [ranks[w]["loop_dense"] for w in range(world)]
    li = [ranks[w]["loop_info"] for w in range(world)]
    for k in range(6):
        late = [w for w in range(world) if w != root and int(li[w][k, 2])]
        want = parts[root].copy()
        for w in [w for w in range(world) if w != root and w not in late] + late:
            want = want + parts[w]
        assert np.array_equal(ranks[root][f"loop_{k}"].view(np.uint32), want.view(np.uint32)), k


@pytest.mark.parametrize("world,root", [(2, 0), (4, 1)])
def test_brick_lists_seeded_walk(tmp_path_factory, cpm, world, root):
    """Ragged grids, 1 and 4 channels, lit sets that jump between a thousandth and nine tenths of the voxels from ticket to ticket (so lists
    outgrow the capacity their count of two tickets before gave them, repeatedly), inexact values, three tickets in flight: after every ticket
    the root's grid is numpy's sum in the library's order -- the root's own, the senders in rank order, the ones that had to send again last
    --, bit for bit; the senders' grids are untouched.  Same workers as above."""
    ranks = _run_lists(tmp_path_factory, world, root)
    sys.path.insert(0, str(REPO / "tests" / "fake_rccl"))
    resent_total = 0
    for fi, (dims, ch) in enumerate((((36, 20, 28), 1), ((17, 33, 12), 4), ((64, 16, 16), 1))):
        infos = [r[f"fuzz_{fi}_info"] for r in ranks]
        for k in range(10):
            parts = [_fuzz_partial(dims, ch, k, w) for w in range(world)]
            late = [w for w in range(world) if w != root and int(infos[w][k, 2])]
            want = parts[root].copy()
            for w in [w for w in range(world) if w != root and w not in late] + late:
                want = want + parts[w]
            assert np.array_equal(ranks[root][f"fuzz_{fi}_{k}"].view(np.uint32), want.view(np.uint32)), (dims, ch, k, late)
            for w in range(world):
                if w != root:
                    assert np.array_equal(ranks[w][f"fuzz_{fi}_{k}"].view(np.uint32), parts[w].view(np.uint32)), (dims, k, w)
                    resent_total += int(infos[w][k, 2])
    assert resent_total >= 3


def _fuzz_partial(dims, ch, k, who):
    dx, dy, dz = dims
    rng = np.random.default_rng(90000 + 1000 * k + 37 * who + dx * dy)
    density = (0.01, 0.3, 0.02, 0.6, 0.6, 0.05, 0.9, 0.001, 0.4, 0.0)[k % 10]
    g = np.zeros((dz, dy, dx, ch), np.float32)
    lit = rng.random((dz, dy, dx)) < density
    if who % 2 == 1:
        lit[: dz // 2] = False
    g[lit] = rng.random((int(lit.sum()), ch), dtype=np.float32) + np.float32(0.01)
    return g.reshape(-1)

"""cpm_trace_order_*: a trace launch whose workgroups take the 256-sample chunks in the order of their measured costs.
The order must not show in the results: photon records, RNG states and light volumes bit for bit those of the default
order, for lattices whose chunk count is and is not a multiple of the 128-chunk blocks the XCD mapping works in, before
and after the order has been re-sorted, with multiple scattering, in progressive mode and through cpm_trace_emitted."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(t):
    return t.contiguous().view(t.torch.int32) if hasattr(t, "torch") else t.contiguous().view(__import__("torch").int32)


def make(ctx, cpm, n_side, adaptive, **kw):
    S, P = cpm.synthetic, cpm.pipeline
    fr = P.PhotonFrame(ctx, S.heterogeneous_volume(64), S.workspace_tf(), n_side, (32,) * 3, light_travel_direction=(0.3, 0.5, -1.0), **kw)
    fr.adaptive_order = adaptive
    return fr


@pytest.mark.parametrize("n_side,kw", [(64, {}), (300, {}), (512, {}), (181, {}), (512, {"max_interactions": 3, "material": (0.3, 0, 0, 0)}),
                                       (256, {"emit_in_tracer": True})])
def test_order_does_not_show_in_the_photons(ctx, cpm, n_side, kw):
    torch = ctx.torch
    ref = make(ctx, cpm, n_side, False, **kw)
    ref.trace()
    want, want_rng = ref.photons.clone(), ref.rng.clone()
    fr = make(ctx, cpm, n_side, True, **kw)
    fr.TRACE_ORDER_EVERY = 2   # measure and re-sort often: default order, measured order, an order measured under the measured order, ...
    for launch in range(5):
        fr.photons.zero_()     # a chunk nobody took would leave zeros behind
        fr.trace()
        torch.cuda.synchronize()
        assert torch.equal(fr.photons.view(torch.int32), want.view(torch.int32)), f"launch {launch}"
        assert torch.equal(fr.rng, want_rng)
    assert fr.trace_order is not None and fr.trace_order.n_light_samples == fr.n


def test_progressive_launches_and_frames(ctx, cpm):
    """RNG write-back (progressive refinement) and the frame's light volume: the same with and without the measured order."""
    torch = ctx.torch
    B = cpm.binding
    a, b = make(ctx, cpm, 384, False), make(ctx, cpm, 384, True)
    b.TRACE_ORDER_EVERY = 1
    for fr in (a, b):
        fr.params.flags |= B.CPM_TRACE_PROGRESSIVE
    for it in range(4):
        va, vb = a.frame_fast().clone(), b.frame_fast().clone()
        assert torch.equal(a.rng, b.rng) and torch.equal(a.photons.view(torch.int32), b.photons.view(torch.int32))
        assert torch.equal(va.view(torch.int32), vb.view(torch.int32))


def test_selected_launches_ignore_the_order_and_sizes_are_checked(ctx, cpm):
    torch = ctx.torch
    B = cpm.binding
    fr = make(ctx, cpm, 256, True)
    fr.trace()
    want = fr.photons.clone()
    order = ctx.trace_order_create(fr.n)
    ctx.trace_set_order(order)
    try:
        idx = torch.arange(0, fr.n, 7, dtype=torch.int32, device=ctx.device)
        fr.photons[idx.long()] = 0
        fr.adaptive_order = False
        fr.trace(recompute_indices=idx, n_recompute=idx.numel())   # PHOTON_RECOMPUTATION launch: index list, not chunks
        torch.cuda.synchronize()
        assert torch.equal(fr.photons.view(torch.int32), want.view(torch.int32))
        other = make(ctx, cpm, 128, False)
        with pytest.raises(B.CpmError, match="another number of samples"):
            other.trace()
    finally:
        ctx.trace_set_order(None)
        order.close()
    with pytest.raises(B.CpmError):
        ctx.trace_order_create(0)
    fresh = ctx.trace_order_create(1000)
    fresh.update()   # nothing measured: the order stays the default
    fresh.close()


def default_order(n_chunks):
    b = np.arange(n_chunks)
    x, j = b & 7, b >> 3
    swz = ((((j >> 4) << 3) + x) << 4) + (j & 15)
    return np.where(b < (n_chunks & ~127), swz, b).astype(np.uint32)


@pytest.mark.parametrize("n_side", [512, 300, 64])
def test_the_table(ctx, cpm, n_side):
    """Before any update: the default mapping.  After: a permutation of the chunks in which every XCD keeps its own chunks
    (whole 4096-sample tiles), its costliest eighth at most comes first, both groups in lattice order, and the costs are cleared."""
    fr = make(ctx, cpm, n_side, False)
    n_chunks = (fr.n + 255) // 256
    order = ctx.trace_order_create(fr.n)
    table, cost, launches = order.read()
    assert np.array_equal(table, default_order(n_chunks)) and not cost.any() and launches == 0
    ctx.trace_set_order(order)
    try:
        fr.trace(); fr.trace()
    finally:
        ctx.trace_set_order(None)
    _, cost, launches = order.read()
    assert launches == 2 and cost.sum() > 0
    order.update()
    table, cleared, launches = order.read()
    assert sorted(table.tolist()) == list(range(n_chunks)) and not cleared.any() and launches == 0
    full = n_chunks & ~127
    assert np.array_equal(table[full:], np.arange(full, n_chunks))
    d = default_order(n_chunks)
    for x in range(8 if full else 0):
        mine, lattice = table[x:full:8], d[x:full:8]
        assert sorted(mine.tolist()) == sorted(lattice.tolist())
        c = cost[mine].astype(np.int64)
        # heavy group = a prefix whose costs all exceed every cost behind it (up to the histogram's resolution: exact here,
        # the costs are far below 2048)
        k = 0
        while k < len(mine) and c[k] > c[k:].min() and c[k] > np.sort(c)[-(len(mine) // 8) - 1]:
            k += 1
        assert k <= len(mine) // 8
        rank = {int(ch): i for i, ch in enumerate(lattice)}
        pos = [rank[int(ch)] for ch in mine]
        assert pos[:k] == sorted(pos[:k]) and pos[k:] == sorted(pos[k:])
        if k:
            assert c[:k].min() > c[k:].max()
    order.close()


def test_any_permutation_of_the_chunks_gives_the_same_photons(ctx, cpm):
    """The measurement hook behind tools/xcd_order_exp.py (cpm_debug_trace_order_write): under an arbitrary permutation of the chunks the
    photons are still the default order's; a table that is not a permutation is refused."""
    torch = ctx.torch
    B = cpm.binding
    ref = make(ctx, cpm, 300, False)
    ref.trace()
    fr = make(ctx, cpm, 300, True)
    fr.trace()                                   # creates the order object (and measures)
    n_chunks = (fr.n + 255) // 256
    rng = np.random.default_rng(5)
    for _ in range(3):
        fr.trace_order.write(rng.permutation(n_chunks).astype(np.uint32))
        fr._traces_since_order = 1               # no re-measure, no re-sort: the written table is what the launch takes
        fr.photons.zero_()
        fr.trace()
        torch.cuda.synchronize()
        assert torch.equal(fr.photons.view(torch.int32), ref.photons.view(torch.int32))
    bad = np.arange(n_chunks, dtype=np.uint32)
    bad[0] = bad[1]
    with pytest.raises(B.CpmError):
        fr.trace_order.write(bad)

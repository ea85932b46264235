"""cpm_trace_lights: several lights' samples in one launch == one cpm_trace per light, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from test_parity_gpu import _n, bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sides,max_inter,dtype", [((64, 48), 1, np.uint8), ((100, 37, 64), 1, np.uint8), ((50, 100), 3, np.uint8),
                                                   ((64, 100), 1, np.float32), ((33,), 1, np.uint16), ((16, 20, 24, 28), 2, np.uint8)])
def test_trace_lights_equals_a_trace_per_light(ctx, cpm, sides, max_inter, dtype):
    """Lights of different directions and ragged sample counts (not multiples of the 256-sample chunk), photons SoA by interaction
    over the sum of the lights: the launch leaves every light's photons at its offset exactly as cpm_trace with that offset does;
    with an order object created for the launch's chunks too, measured and re-sorted in between."""
    S, P, B = cpm.synthetic, cpm.pipeline, cpm.binding
    torch = ctx.torch
    vol_np = S.heterogeneous_volume(48)
    if dtype == np.uint16:
        vol_np = (vol_np.astype(np.uint16) * 257).astype(np.uint16)
    elif dtype == np.float32:
        vol_np = (vol_np.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    vol = ctx.volume_create(vol_np)
    tf = ctx.tf_create(S.workspace_tf())
    dirs = [(0.3, 0.5, -1.0), (-0.4, 0.2, -1.0), (0.1, -0.6, 1.0), (1.0, 0.2, 0.1)]
    frames = [P.PhotonFrame(ctx, vol, tf, s, (16, 16, 16), light_travel_direction=dirs[k], max_interactions=max_inter, material=(0.3, 0, 0, 0), seed=k)
              for k, s in enumerate(sides)]
    ns = [f.n for f in frames]
    N = sum(ns)
    offs = np.concatenate([[0], np.cumsum(ns)[:-1]]).astype(int).tolist()
    rng = torch.cat([f.rng_initial for f in frames]).contiguous()
    params = B.TraceParams()
    C.memmove(C.byref(params), C.byref(frames[0].params), C.sizeof(params))
    params.total_photons = N
    want = torch.full((N * max_inter, 8), -7.0, dtype=torch.float32, device=ctx.device)
    for f, n, off in zip(frames, ns, offs):
        params.photon_offset, params.n_light_samples = off, n
        ctx.trace(vol, tf, f.aabb, params, f.light_samples, f.isect, rng, want)
    spans = ctx.light_spans([(f.light_samples, f.isect, n, off) for f, n, off in zip(frames, ns, offs)])
    assert ctx.trace_lights_order_samples(spans) == 256 * sum((n + 255) // 256 for n in ns)
    params.photon_offset, params.n_light_samples = 12345, -1     # ignored by the launch
    got = torch.full((N * max_inter, 8), -7.0, dtype=torch.float32, device=ctx.device)
    ctx.trace_lights(vol, tf, frames[0].aabb, params, spans, rng, got)
    assert np.array_equal(bits(_n(got)), bits(_n(want)))
    assert (_n(got)[:, 0] < 1e30).any()
    # under an order object for the launch's chunks: measured, re-sorted, traced again
    order = ctx.trace_order_create(ctx.trace_lights_order_samples(spans))
    for measure in (True, False, False):
        ctx.trace_set_order(order, measure)
        got.fill_(-7.0)
        ctx.trace_lights(vol, tf, frames[0].aabb, params, spans, rng, got)
        ctx.trace_set_order(None)
        if measure:
            order.update()
        assert np.array_equal(bits(_n(got)), bits(_n(want)))
    order.close()
    # refusals: no lights, too many, a light beyond the photon array
    with pytest.raises(B.CpmError):
        ctx.trace_lights(vol, tf, frames[0].aabb, params, ctx.light_spans([]), rng, got)
    with pytest.raises(B.CpmError):
        ctx.trace_lights(vol, tf, frames[0].aabb, params, ctx.light_spans([(frames[0].light_samples, frames[0].isect, ns[0], 0)] * 5), rng, got)
    with pytest.raises(B.CpmError):
        ctx.trace_lights(vol, tf, frames[0].aabb, params, ctx.light_spans([(frames[0].light_samples, frames[0].isect, ns[0], N)]), rng, got)


@pytest.mark.parametrize("sides,max_inter", [((96, 70), 1), ((64, 50, 40), 2)])
def test_importance_retrace_lights_equals_a_call_per_light(ctx, cpm, sides, max_inter):
    """cpm_photon_importance_retrace_lights: the lights' tiles in one launch -- photons, replaced records, importance keys, index list and
    count are those of one cpm_photon_importance_retrace per light in the same order, bit for bit; repeated selections (the tile order
    is measured and re-sorted in between) keep giving them."""
    S, P, B = cpm.synthetic, cpm.pipeline, cpm.binding
    torch = ctx.torch
    UNCHANGED = np.uint32(0x7fffffff)
    vdim, region = 64, 8
    base = [(0.0, 1, 1, 1, 0.0), (0.45, 1, 0.5, 0.2, 0.0), (0.55, 0.6, 0.3, 0.1, 0.05), (0.8, 0.9, 0.2, 0.3, 0.4), (1.0, 0.1, 0.6, 0.7, 0.5)]
    edit = list(base)
    edit[3] = (0.85,) + base[3][1:]
    vol = ctx.volume_create(S.heterogeneous_volume(vdim))
    tf = ctx.tf_create(S.tf_from_points(base))
    dirs = [(0.3, 0.5, -1.0), (-0.4, 0.2, -1.0), (0.1, -0.6, 1.0)]
    frames = [P.PhotonFrame(ctx, vol, tf, s, (16, 16, 16), light_travel_direction=dirs[k], max_interactions=max_inter, material=(0.3, 0, 0, 0), seed=k)
              for k, s in enumerate(sides)]
    ns = [f.n for f in frames]
    N = sum(ns)
    offs = np.concatenate([[0], np.cumsum(ns)[:-1]]).astype(int).tolist()
    rng = torch.cat([f.rng_initial for f in frames]).contiguous()
    params = B.TraceParams()
    C.memmove(C.byref(params), C.byref(frames[0].params), C.sizeof(params))
    params.total_photons = N
    spans = ctx.light_spans([(f.light_samples, f.isect, n, off) for f, n, off in zip(frames, ns, offs)])
    before = torch.zeros((N * max_inter, 8), dtype=torch.float32, device=ctx.device)
    ctx.trace_lights(vol, tf, frames[0].aabb, params, spans, rng, before)
    tf.update(S.tf_from_points(edit))
    gd = (vdim // region,) * 3
    r = np.random.default_rng(5)
    grid = r.random(gd[0] * gd[1] * gd[2], dtype=np.float32)
    grid[r.random(grid.size) < 0.9] = 0
    dgrid = torch.from_numpy(grid).to(ctx.device)
    t2i = list(vol.desc.texture_to_index)
    imp0 = torch.from_numpy(np.full(N, UNCHANGED, np.uint32).view(np.int32)).to(ctx.device)

    def run(sel, batched):
        imp, idx = imp0.clone(), torch.full((N,), -1, dtype=torch.int32, device=ctx.device)
        ph, old = before.clone(), torch.full((max_inter * N, 8), 7.0, dtype=torch.float32, device=ctx.device)
        sel.begin()
        if batched:
            sel.photon_importance_retrace_lights(dgrid, gd, (float(region),) * 3, t2i, vol, tf, frames[0].aabb, params, spans, imp, rng, ph, old)
        else:
            for f, n, off in zip(frames, ns, offs):
                params.photon_offset, params.n_light_samples = off, n
                sel.photon_importance_retrace(dgrid, gd, (float(region),) * 3, t2i, vol, tf, f.aabb, params, f.light_samples, f.isect, imp, rng, ph, old)
        sel.finish(idx)
        cnt = sel.count()
        torch.cuda.synchronize()
        return cnt, _n(idx), bits(_n(ph)), bits(_n(old)), _n(imp)

    a, b = ctx.selection_create(N), ctx.selection_create(N)
    want = run(a, False)
    assert 0 < want[0] < N and (want[2] != bits(_n(before))).any()
    for _ in range(3):      # (the first launch of a selection is measured: later ones run in the re-sorted tile order)
        got = run(b, True)
        assert got[0] == want[0]
        for g, w in zip(got[1:], want[1:]):
            assert np.array_equal(g, w)
    a.close(); b.close()

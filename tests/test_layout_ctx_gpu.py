"""Photon records in two planes give the same bits as float8 records -- every entry point that reads or writes an N * I record buffer,
driven directly and through the pipeline's correlated mapper (full frame, TF edit, the three forms of the update) -- in both ways of
saying so: PER BUFFER (cpm_records_describe: the buffer's layout and its own N * I, on a context whose default stays float8; no
context-wide mode anywhere) and as a context's default (cpm_set_photon_layout, the convenience a described buffer overrides)."""
import numpy as np
import pytest

from test_parity_gpu import _t, _n, bits

pytestmark = pytest.mark.gpu
FLT_MAX = np.float32(3.402823466e+38)


@pytest.fixture()
def planar_ctx(cpm):
    c = cpm.binding.Context(0)
    c.set_photon_layout(cpm.binding.CPM_PHOTONS_PLANAR)
    assert c.photon_layout() == cpm.binding.CPM_PHOTONS_PLANAR
    yield c
    c.close()


def to_planes(rec8):
    """float8 rows -> the two-plane buffer (same shape, other addresses)."""
    n = rec8.shape[0]
    out = np.empty_like(rec8)
    flat = out.reshape(-1)
    flat[: 4 * n] = rec8[:, :4].reshape(-1)
    flat[4 * n:] = rec8[:, 4:].reshape(-1)
    return out


def from_planes(buf):
    n = buf.shape[0]
    flat = buf.reshape(-1)
    return np.concatenate([flat[: 4 * n].reshape(n, 4), flat[4 * n:].reshape(n, 4)], axis=1)


def random_records(rng, n, inter, sentinels=0.1):
    ph = np.zeros((n * inter, 8), np.float32)
    ph[:, :3] = rng.random((n * inter, 3), dtype=np.float32)
    ph[:, 3:6] = rng.random((n * inter, 3), dtype=np.float32) + np.float32(0.05)
    ph[:, 6:] = rng.random((n * inter, 2), dtype=np.float32) * np.float32(3.0)
    dead = rng.random(n * inter) < sentinels
    ph[dead, :3] = FLT_MAX
    return ph


def test_layout_is_validated(cpm, planar_ctx):
    with pytest.raises(cpm.binding.CpmError):
        planar_ctx.set_photon_layout(7)
    assert planar_ctx.photon_layout() == cpm.binding.CPM_PHOTONS_PLANAR


@pytest.mark.parametrize("channels", [1, 4])
def test_record_readers_agree_between_layouts(ctx, planar_ctx, cpm, channels):
    """splat (records), splat_selected, copy_indexed, snapshot_selected, mark_touched_bricks, bin + gather, bin_fast + gather_fast."""
    B = cpm.binding
    torch = ctx.torch
    rng = np.random.default_rng(41 + channels)
    n, inter, dims = 6000, 2, (24, 20, 16)
    ph = random_records(rng, n, inter)
    idx = np.sort(rng.choice(n, 700, replace=False)).astype(np.uint32)
    grid = B.default_grid_desc(dims, channels)
    cells = dims[0] * dims[1] * dims[2]
    radius = float(np.float32(1.4) / np.float32(max(dims)))
    scale = float(B.relative_irradiance_scale(radius, float(n)))
    res = {}
    for name, c, buf in (("float8", ctx, ph), ("planes", planar_ctx, to_planes(ph)), ("described", ctx, to_planes(ph))):
        d_ph, d_idx = _t(c, buf), _t(c, idx)
        if name == "described":       # the default context, the buffer itself described as two planes of n * inter records
            c.records_describe(d_ph, B.CPM_PHOTONS_PLANAR, n * inter)
        r = {}
        out = torch.zeros(cells * channels, dtype=torch.float32, device=c.device)
        c.splat_records(d_ph, n * inter, n, grid, radius, scale, out)
        r["splat"] = _n(out)
        out = torch.zeros(cells * channels, dtype=torch.float32, device=c.device)
        c.splat_selected(d_ph, d_idx, idx.size, grid, radius, scale, -1.0, n, inter, out)
        r["splat_selected"] = _n(out)
        aligned = torch.zeros((idx.size * inter + 3, 8), dtype=torch.float32, device=c.device)
        c.copy_indexed_photons(d_ph, d_idx, idx.size, 0.5, n, inter, aligned, out_offset=3)
        r["aligned"] = _n(aligned)                                   # compact copies keep the float8 record in either context
        snap = torch.full((n * inter, 8), 5.0, dtype=torch.float32, device=c.device)
        c.snapshot_selected_photons(d_ph, d_idx, idx.size, n, inter, snap)     # (the snapshot is laid out like the buffer it is taken from)
        r["snapshot"] = _n(snap) if name == "float8" else from_planes(_n(snap))
        mask = torch.zeros(((dims[0] + 3) // 4) * ((dims[1] + 3) // 4) * ((dims[2] + 3) // 4), dtype=torch.uint8, device=c.device)
        c.mark_touched_bricks(d_ph, d_idx, idx.size, n, inter, grid, radius, mask)
        r["mask"] = _n(mask)
        m = n * inter
        order = torch.zeros(m, dtype=torch.int32, device=c.device)
        cell_start = torch.zeros(cells + 1, dtype=torch.int32, device=c.device)
        srt = torch.zeros((m, 4 if channels == 1 else 8), dtype=torch.float32, device=c.device)
        c.bin(d_ph, m, grid, order, cell_start, srt)
        out = torch.zeros(cells * channels, dtype=torch.float32, device=c.device)
        c.gather(srt, cell_start, m, grid, radius, scale, out)
        r["order"], r["cell_start"], r["sorted"], r["gather"] = _n(order), _n(cell_start), _n(srt), _n(out)
        table = torch.zeros(max(c.fast_table_entries(grid, m), 1), dtype=torch.int32, device=c.device)
        srt2 = torch.zeros((max(c.fast_record_capacity(grid, m, radius), 1), 4 if channels == 1 else 8), dtype=torch.float32, device=c.device)
        c.bin_fast(d_ph, m, grid, radius, table, srt2)                # (no layout given: the context's)
        out = torch.zeros(cells * channels, dtype=torch.float32, device=c.device)
        c.gather_fast(srt2, table, m, grid, radius, scale, out)
        r["gather_fast"] = _n(out)
        # a call over the FIRST n records of the n * inter in the buffer (interaction 0 alone): a described buffer's plane B stays n * inter
        # float4 behind plane A; a context's default takes the call's own count for that distance, so only float8 and described can do this
        if name != "planes":
            c.bin_fast(d_ph, n, grid, radius, table, srt2)
            out = torch.zeros(cells * channels, dtype=torch.float32, device=c.device)
            c.gather_fast(srt2, table, n, grid, radius, scale, out)
            r["gather_fast_prefix"] = _n(out)
            c.bin(d_ph, n, grid, order, cell_start, srt)
            out = torch.zeros(cells * channels, dtype=torch.float32, device=c.device)
            c.gather(srt, cell_start, n, grid, radius, scale, out)
            r["gather_prefix"] = _n(out)
        if name == "described":
            c.records_forget(d_ph)
            with pytest.raises(B.CpmError):
                c.records_describe(d_ph, 9, n * inter)
        res[name] = r
    a = res["float8"]
    for b in (res["planes"], res["described"]):
        for k in ("aligned", "snapshot", "mask", "order", "cell_start", "sorted", "gather", "gather_fast"):
            assert np.array_equal(bits(a[k]), bits(b[k])), k
        for k in ("splat", "splat_selected"):                             # float atomics: the order of the adds is not defined
            np.testing.assert_allclose(b[k], a[k], rtol=1e-4, atol=1e-6 * float(np.abs(a[k]).max()))
    for k in ("gather_fast_prefix", "gather_prefix"):
        assert np.array_equal(bits(a[k]), bits(res["described"][k])), k
    assert channels == 4 or not np.array_equal(bits(a["gather_fast_prefix"]), bits(a["gather_fast"]))   # (the second interaction's records do count)
    assert a["gather"].any() and a["mask"].any() and (a["snapshot"] != 5.0).any()


@pytest.mark.parametrize("way", ["buffers", "context"])
@pytest.mark.parametrize("max_inter,form", [(1, "retrace_in_pass"), (2, "retrace_in_pass"), (1, "select_then_trace"), (2, "legacy")])
def test_correlated_mapper_in_a_planar_context(ctx, planar_ctx, cpm, max_inter, form, way):
    """Full frame, TF edits and updates with the records in two planes: the same records (converted back), index lists and importance
    keys bit for bit, the light volume bit for bit after a full frame and within the splat tolerance after a delta update."""
    S, P = cpm.synthetic, cpm.pipeline
    vol_np = S.heterogeneous_volume(64)
    base = [(0.0, 1, 1, 1, 0.0), (0.45, 1, 0.5, 0.2, 0.0), (0.55, 0.6, 0.3, 0.1, 0.05), (0.8, 0.9, 0.2, 0.3, 0.4), (1.0, 0.1, 0.6, 0.7, 0.5)]
    edit = list(base)
    edit[3] = (0.85,) + base[3][1:]
    kw = dict(light_travel_direction=(0.3, 0.5, -1.0), tf_points=base, max_interactions=max_inter, material=(0.3, 0, 0, 0),
              incremental_threshold_percent=100.0)
    pair = []
    for c, how in ((ctx, None), (planar_ctx, None) if way == "context" else (ctx, "planar")):
        # (way == "buffers": the SAME default context as the float8 mapper; its record buffers are described one by one)
        m = P.CorrelatedPhotonMapper(c, vol_np, S.tf_from_points(base), 160, (32, 32, 32), records_layout=how, **kw)
        if form == "select_then_trace":
            m.retrace_in_importance_pass = False
        if form == "legacy":
            m.fused = False
        pair.append(m)
    a, b = pair
    assert b.photon_layout == cpm.binding.CPM_PHOTONS_PLANAR and a.photon_layout == cpm.binding.CPM_PHOTONS_INTERLEAVED
    assert way == "context" or b.ctx.photon_layout() == cpm.binding.CPM_PHOTONS_INTERLEAVED
    la, lb = _n(a.full_frame()), _n(b.full_frame())
    assert np.array_equal(bits(la), bits(lb)) and la.any()
    assert np.array_equal(bits(_n(a.photons)), bits(_n(b.records())))
    assert not np.array_equal(bits(_n(a.photons)), bits(_n(b.photons)))        # (the planar buffer really is laid out differently)
    for pts in (edit, base):
        a.set_transfer_function(pts); b.set_transfer_function(pts)
        na, nb = a.correlated_update(), b.correlated_update()
        assert na == nb > 0 and a.last_path == b.last_path == "incremental"
        assert np.array_equal(bits(_n(a.photons)), bits(_n(b.records())))
        assert np.array_equal(_n(a.indices, np.uint32)[:na], _n(b.indices, np.uint32)[:nb])
        assert np.array_equal(_n(a.importance, np.uint32), _n(b.importance, np.uint32))
        la, lb = _n(a.light_volume), _n(b.light_volume)
        np.testing.assert_allclose(lb, la, rtol=1e-3, atol=2e-5 * float(la.max()))
    # a rebuild after the edits: bit for bit again
    a._full_light_volume(); b._full_light_volume()
    assert np.array_equal(bits(_n(a.light_volume)), bits(_n(b.light_volume)))
    b.forget_described()


def test_call_flag_against_context_layout_is_refused(ctx, cpm):
    """The importance pass reads the records in the CONTEXT's layout: a planar flag on a call of an interleaved context cannot be served."""
    S, P = cpm.synthetic, cpm.pipeline
    B = cpm.binding
    base = [(0.0, 1, 1, 1, 0.0), (0.6, 1, 1, 1, 0.0), (1.0, 1, 1, 1, 0.4)]
    m = P.CorrelatedPhotonMapper(ctx, S.heterogeneous_volume(32), S.tf_from_points(base), 64, (16, 16, 16), light_travel_direction=(0.3, 0.5, -1.0),
                                 tf_points=base)
    m.full_frame()
    m.set_transfer_function([(0.0, 1, 1, 1, 0.0), (0.5, 1, 1, 1, 0.0), (1.0, 1, 1, 1, 0.4)])
    sel = ctx.selection_create(m.n)
    old = ctx.torch.empty((m.n, 8), dtype=ctx.torch.float32, device=ctx.device)
    sel.begin()
    m.params.flags = B.CPM_TRACE_PHOTONS_PLANAR
    with pytest.raises(B.CpmError):
        sel.photon_importance_retrace(m.importance_grid, m.brick_dims, (float(m.region),) * 3, list(m.vol.desc.texture_to_index), m.vol, m.tf, m.aabb,
                                      m.params, m.light_samples, m.isect, m.importance, m.rng, m.photons, old)
    sel.close()

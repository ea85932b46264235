"""The two-plane photon record layout (CPM_TRACE_PHOTONS_PLANAR / cpm_bin_fast_layout / cpm_photons_convert, include/cpm/cpm.h).

A layout option, not another computation: the tracer stores the same two 16-byte halves of every record (ref
progressivephotonmapping/cl/photon.cl:49-63, id = offset + k N + i as cl/photontracer.cl:166) at other addresses, the brick bin
reads position + first power channel from one plane.  So everything is compared bit for bit: records after conversion against the
float8 trace (which the parity tests hold to the oracle), the brick table against the float8 bin's, the light volume against the
float8 frame's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frame(cpm, ctx, vdim, nside, gdim, interactions=1, channels=1, **kw):
    S, P = cpm.synthetic, cpm.pipeline
    return P.PhotonFrame(ctx, S.heterogeneous_volume(vdim), S.workspace_tf(), nside, (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0),
                         max_interactions=interactions, channels=channels, **kw)


def _u32(t):
    return t.detach().cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("vdim,nside,gdim,interactions,channels", [(64, 96, 32, 1, 1), (64, 96, 32, 3, 1), (48, 131, 24, 1, 4), (64, 64, 32, 4, 4)])
def test_planar_trace_bin_gather_equal_the_float8_frame(cpm, ctx, vdim, nside, gdim, interactions, channels):
    B = cpm.binding
    fr = _frame(cpm, ctx, vdim, nside, gdim, interactions, channels)
    n = fr.n * fr.I
    fr.rng.copy_(fr.rng_initial)
    fr.trace(); fr.bin_fast(); fr.gather_fast()
    want_records, want_table, want_lv = fr.photons.clone(), fr.brick_table.clone(), fr.light_volume.clone()
    assert float(want_lv.sum()) > 0

    fr.set_planar_records(True)
    fr.rng.copy_(fr.rng_initial)
    fr.photons.fill_(-7.0)
    fr.light_volume.fill_(-1.0)
    fr.trace(); fr.bin_fast(); fr.gather_fast()
    ctx.torch.cuda.synchronize()
    # the planes: A = (x, y, z, powerR) of record j at row j of the first half, B behind it
    planes = fr.photons.reshape(-1).reshape(2, n, 4)
    assert np.array_equal(_u32(planes[0]), _u32(want_records[:, :4])) and np.array_equal(_u32(planes[1]), _u32(want_records[:, 4:]))
    assert np.array_equal(_u32(fr.records()), _u32(want_records))
    assert np.array_equal(_u32(fr.brick_table), _u32(want_table))
    assert np.array_equal(_u32(fr.light_volume), _u32(want_lv))
    # and back: interleaved -> planar is the inverse
    again = ctx.torch.empty_like(fr.photons)
    ctx.photons_convert(want_records, B.CPM_PHOTONS_INTERLEAVED, again, B.CPM_PHOTONS_PLANAR, n)
    ctx.torch.cuda.synchronize()
    assert np.array_equal(_u32(again), _u32(fr.photons))


def test_planar_with_the_emitting_tracer_and_a_shard(cpm, ctx):
    """cpm_trace_emitted (samples evaluated in the kernel) and a rank's tile shard (photon_indices) write planar records too."""
    sharding = __import__(cpm.__name__ + ".sharding", fromlist=["x"])
    for kw in ({"emit_in_tracer": True}, {"photon_indices": sharding.shard_tiles(160 * 160, 1, 3)}):
        fr = _frame(cpm, ctx, 64, 160, 32, **kw)
        fr.rng.copy_(fr.rng_initial)
        fr.trace(); fr.bin_fast(); fr.gather_fast()
        want_records, want_lv = fr.photons.clone(), fr.light_volume.clone()
        fr.set_planar_records(True)
        fr.rng.copy_(fr.rng_initial)
        fr.trace(); fr.bin_fast(); fr.gather_fast()
        ctx.torch.cuda.synchronize()
        assert np.array_equal(_u32(fr.records()), _u32(want_records)) and np.array_equal(_u32(fr.light_volume), _u32(want_lv))


def test_planar_argument_errors(cpm, ctx):
    B = cpm.binding
    fr = _frame(cpm, ctx, 32, 32, 16)
    fr.trace()
    other = ctx.torch.empty_like(fr.photons)
    with pytest.raises(B.CpmError) as e:
        ctx.photons_convert(fr.photons, 2, other, B.CPM_PHOTONS_PLANAR, fr.n)
    assert e.value.status == -1
    with pytest.raises(B.CpmError) as e:
        ctx.photons_convert(fr.photons, B.CPM_PHOTONS_INTERLEAVED, fr.photons, B.CPM_PHOTONS_PLANAR, fr.n)  # aliased
    assert e.value.status == -1
    ctx.photons_convert(fr.photons, B.CPM_PHOTONS_INTERLEAVED, other, B.CPM_PHOTONS_PLANAR, 0)  # nothing to do
    fr.bin_fast()
    with pytest.raises(B.CpmError) as e:
        ctx.bin_fast(fr.photons, fr.n, fr.grid, fr.radius, fr.brick_table, fr.sorted_fast, layout=3)
    assert e.value.status == -1


def test_planar_config2_full_size(cpm, ctx):
    """BASELINE config 2 (256^3, 1 048 576 photons, 128^3 light volume): the planar frame is the float8 frame, every word."""
    fr = _frame(cpm, ctx, 256, 1024, 128)
    fr.rng.copy_(fr.rng_initial)
    fr.trace(); fr.bin_fast(); fr.gather_fast()
    want_records, want_table, want_lv = fr.photons.clone(), fr.brick_table.clone(), fr.light_volume.clone()
    fr.set_planar_records(True)
    fr.rng.copy_(fr.rng_initial)
    fr.trace(); fr.bin_fast(); fr.gather_fast()
    ctx.torch.cuda.synchronize()
    assert np.array_equal(_u32(fr.records()), _u32(want_records))
    assert np.array_equal(_u32(fr.brick_table), _u32(want_table))
    assert np.array_equal(_u32(fr.light_volume), _u32(want_lv))


def test_planar_trace_lights_and_brick_mask_or(cpm, ctx):
    """cpm_trace_lights (several lights, one launch) writes the two-plane layout too; and cpm_brick_mask_or is the union of two masks."""
    import ctypes as C
    S, P, B = cpm.synthetic, cpm.pipeline, cpm.binding
    torch = ctx.torch
    vol = ctx.volume_create(S.heterogeneous_volume(48))
    tf = ctx.tf_create(S.workspace_tf())
    frames = [P.PhotonFrame(ctx, vol, tf, s, (16, 16, 16), light_travel_direction=d, max_interactions=2, material=(0.3, 0, 0, 0), seed=k)
              for k, (s, d) in enumerate(((50, (0.3, 0.5, -1.0)), (37, (-0.4, 0.2, -1.0))))]
    ns = [f.n for f in frames]
    N, I = sum(ns), 2
    rng = torch.cat([f.rng_initial for f in frames]).contiguous()
    params = B.TraceParams()
    C.memmove(C.byref(params), C.byref(frames[0].params), C.sizeof(params))
    params.total_photons = N
    spans = ctx.light_spans([(f.light_samples, f.isect, n, off) for f, n, off in zip(frames, ns, (0, ns[0]))])
    want = torch.full((N * I, 8), -7.0, dtype=torch.float32, device=ctx.device)
    ctx.trace_lights(vol, tf, frames[0].aabb, params, spans, rng.clone(), want)
    params.flags |= B.CPM_TRACE_PHOTONS_PLANAR
    planar = torch.full((N * I, 8), -7.0, dtype=torch.float32, device=ctx.device)
    ctx.trace_lights(vol, tf, frames[0].aabb, params, spans, rng.clone(), planar)
    back = torch.empty_like(planar)
    ctx.photons_convert(planar, B.CPM_PHOTONS_PLANAR, back, B.CPM_PHOTONS_INTERLEAVED, N * I)
    torch.cuda.synchronize()
    assert np.array_equal(_u32(back), _u32(want)) and (want[:, 0] < 1e30).any()
    # the importance pass reads float8 records: a planar re-trace through it is refused, not misread
    g = torch.Generator(device="cpu").manual_seed(3)
    a = (torch.rand(1000, generator=g) < 0.3).to(torch.uint8).to(ctx.device) * 7     # (any non-zero value counts)
    b = (torch.rand(1000, generator=g) < 0.3).to(torch.uint8).to(ctx.device)
    want_or = ((a != 0) | (b != 0)).to(torch.uint8)
    ctx.brick_mask_or(a, b)
    torch.cuda.synchronize()
    assert torch.equal(a, want_or)
    ctx.brick_mask_or(a, b, n=0)   # nothing to do

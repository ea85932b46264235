"""Temporal interpolation between time steps (mixKernel / volume_mix.frag) against the oracle, on the GPU."""
import numpy as np
import pytest

from test_parity_gpu import _t, _n, bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [0, 1, 3, 4, 1023, 262144 + 5])
@pytest.mark.parametrize("a", [0.0, 0.37, 1.0])
def test_mix_f32(ctx, oracle, n, a):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) * 10).astype(np.float32)
    y = (rng.standard_normal(n) * 10).astype(np.float32)
    xd, yd = _t(ctx, x), _t(ctx, y)
    out = ctx.torch.full((n,), -1.0, dtype=ctx.torch.float32, device=ctx.device)
    ctx.mix_buffers(xd, yd, a, out)
    assert np.array_equal(bits(_n(out)), bits(oracle.mix_f32(x, y, a)))
    if a == 0.0:
        assert np.array_equal(_n(out), x)  # x + (y - x) * 0


@pytest.mark.parametrize("n_pairs", [1, 5, 4096, 32 * 32 * 32 + 3])
@pytest.mark.parametrize("a", [0.0, 0.25, 0.999, 1.0])
def test_mix_minmax_grid(ctx, oracle, n_pairs, a):
    """MinMaxUniformGrid3D (Vec2UINT16) through convert_float2 / convert_ushort2 (round toward zero)."""
    rng = np.random.default_rng(n_pairs)
    x = rng.integers(0, 65536, (n_pairs, 2)).astype(np.uint16)
    y = rng.integers(0, 65536, (n_pairs, 2)).astype(np.uint16)
    x[0], y[0] = (0, 65535), (65535, 0)
    xd = _t(ctx, x.view(np.int16))
    yd = _t(ctx, y.view(np.int16))
    out = ctx.torch.zeros((n_pairs, 2), dtype=ctx.torch.int16, device=ctx.device)
    ctx.mix_buffers(xd, yd, a, out)
    want = oracle.mix_u16x2(x, y, a)
    assert np.array_equal(_n(out, np.uint16), want)
    if a == 0.0:
        assert np.array_equal(want, x)


@pytest.mark.parametrize("shape,dtype", [((32, 32, 32), np.uint8), ((7, 9, 11), np.uint8), ((16, 20, 24), np.uint16),
                                          ((12, 12, 12), np.float32)])
@pytest.mark.parametrize("w", [0.0, 0.5, 0.3125, 1.0])
def test_volume_mix(ctx, oracle, shape, dtype, w):
    rng = np.random.default_rng(sum(shape))
    if dtype == np.float32:
        a, b = rng.random(shape, dtype=np.float32), rng.random(shape, dtype=np.float32)
    else:
        hi = np.iinfo(dtype).max + 1
        a, b = rng.integers(0, hi, shape).astype(dtype), rng.integers(0, hi, shape).astype(dtype)
    va, vb, vo = ctx.volume_create(a), ctx.volume_create(b), ctx.volume_create(np.zeros_like(a))
    ctx.volume_mix(va, vb, w, vo)
    got = vo.download()
    want = oracle.volume_mix(oracle.volume(a), oracle.volume(b), w, a)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    if w == 0.0 and dtype != np.float32:
        assert np.array_equal(got, a)  # v/255 * 1 + y * 0 -> rint(v/255 * 255) == v
    if w == 1.0 and dtype != np.float32:
        assert np.array_equal(got, b)


def test_volume_mix_full_size_properties(ctx, cpm):
    """Config 5 size (256^3 u8): the mix is monotone in w voxel by voxel and stays between its inputs."""
    S = cpm.synthetic
    a = S.heterogeneous_volume(256, S.sequence_blob_center(3))
    b = S.heterogeneous_volume(256, S.sequence_blob_center(4))
    va, vb, vo = ctx.volume_create(a), ctx.volume_create(b), ctx.volume_create(np.zeros_like(a))
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    prev = None
    for w in (0.0, 0.25, 0.5, 0.75, 1.0):
        ctx.volume_mix(va, vb, w, vo)
        got = vo.download()
        assert (got >= lo).all() and (got <= hi).all()
        if prev is not None:
            d0 = got.astype(np.int16) - prev.astype(np.int16)
            sign = np.sign(b.astype(np.int16) - a.astype(np.int16))
            assert (d0 * sign >= 0).all()
        prev = got
    assert np.array_equal(prev, b)


def test_mix_errors(ctx, cpm):
    t = ctx.torch
    x = t.zeros(16, dtype=t.float32, device=ctx.device)
    with pytest.raises(cpm.binding.CpmError):
        ctx.mix_buffers(x, x, 0.5, x, kind=7)
    with pytest.raises(cpm.binding.CpmError):
        ctx.mix_buffers(x[1:], x[1:], 0.5, x[1:])  # not 16-byte aligned
    a = ctx.volume_create(np.zeros((4, 4, 4), np.uint8))
    b = ctx.volume_create(np.zeros((4, 4, 8), np.uint8))
    with pytest.raises(cpm.binding.CpmError):
        ctx.volume_mix(a, b, 0.5, a)


@pytest.mark.parametrize("shape,dtype", [((32, 32, 32), np.uint8), ((24, 20, 16), np.uint16), ((12, 16, 20), np.float32), ((9, 7, 11), np.uint8)])
def test_volume_mix_leaves_the_tracers_copy_in_step(ctx, oracle, cpm, shape, dtype):
    """cpm_volume_mix marks the footprint-ordered copy the tracer reads as stale and the next trace over all the samples re-derives it
    (row lengths on and off the 4-byte grid): a trace through the mixed volume gives the photons of a trace through a volume
    created from the mixed voxels."""
    S, P = cpm.synthetic, cpm.pipeline
    rng = np.random.default_rng(sum(shape) + 1)
    if dtype == np.float32:
        a, b = rng.random(shape, dtype=np.float32), rng.random(shape, dtype=np.float32)
    else:
        hi = np.iinfo(dtype).max + 1
        a, b = rng.integers(0, hi, shape).astype(dtype), rng.integers(0, hi, shape).astype(dtype)
    va, vb, vo = ctx.volume_create(a), ctx.volume_create(b), ctx.volume_create(np.zeros_like(a))
    ctx.volume_mix(va, vb, 0.3125, vo)
    mixed = vo.download()
    fresh = ctx.volume_create(mixed)
    tf = S.homogeneous_tf(0.1)
    f1 = P.PhotonFrame(ctx, vo, tf, 96, (16, 16, 16), light_travel_direction=(0.3, 0.5, -1.0))
    f2 = P.PhotonFrame(ctx, fresh, tf, 96, (16, 16, 16), light_travel_direction=(0.3, 0.5, -1.0))
    f1.trace(); f2.trace()
    ctx.torch.cuda.synchronize()
    assert ctx.torch.equal(f1.photons.view(ctx.torch.int32), f2.photons.view(ctx.torch.int32))
    assert (f1.photons[:, 0] < 1e30).any()


@pytest.mark.parametrize("path", ["one launch", "selected", "launch by launch"])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_updates_through_a_mixed_volume_read_the_linear_block(ctx, cpm, path, dtype):
    """cpm_volume_mix leaves the tracer's footprint copy stale: the correlated update's re-traces (one-launch importance + re-trace,
    cpm_trace_selected over the device count, cpm_trace over a host-counted index list) then sample the volume's linear block --
    four x-pair fetches instead of one footprint fetch, the same eight voxels and lerps -- and a trace over all the samples
    re-derives the copy first.  Photons, selections and importance grids are those of a mapper stepping through volumes
    created from the mixed voxels (copies in step from the start), bit for bit."""
    S, P = cpm.synthetic, cpm.pipeline
    tfp = [(0.0, 1, 1, 1, 0.0), (0.55, 1, 0.5, 0.2, 0.0), (0.7, 0.6, 0.3, 0.1, 0.3), (1.0, 0.1, 0.6, 0.7, 0.6)]

    def as_type(v):
        if dtype == np.uint8:
            return v
        if dtype == np.uint16:
            return (v.astype(np.uint16) * 257).astype(np.uint16)
        return (v.astype(np.float32) / np.float32(255.0)).astype(np.float32)

    keys = [as_type(S.heterogeneous_volume(64, S.sequence_blob_center(t * 10, 32))) for t in range(3)]
    kw = dict(light_travel_direction=(0.3, 0.5, -1.0), tf_points=tfp, incremental_threshold_percent=100.0)
    if path == "launch by launch":
        kw["fused"] = False
    key_vols = [ctx.volume_create(v) for v in keys]
    mixed_out = [ctx.volume_create(np.zeros_like(keys[0])) for _ in range(2)]   # the player's two output volumes, used in turn

    def make(first):
        m = P.CorrelatedPhotonMapper(ctx, first, S.tf_from_points(tfp), 128, (32, 32, 32), **{k: v for k, v in kw.items() if k != "fused"})
        m.fused = kw.get("fused", True)
        m.retrace_in_importance_pass = path == "one launch"
        return m

    lazy, eager = make(key_vols[0]), make(ctx.volume_create(keys[0]))
    lazy.full_frame(); eager.full_frame()
    assert np.array_equal(bits(_n(lazy.photons)), bits(_n(eager.photons)))
    retraced = []
    for k, (i, w) in enumerate([(0, 0.5), (1, 0.0), (1, 0.75), (1, 1.0)]):
        out = mixed_out[k % 2]
        ctx.volume_mix(key_vols[i], key_vols[i + 1], w, out)
        fresh = ctx.volume_create(out.download())            # the same voxels, its copy in step
        lazy.set_volume(out); eager.set_volume(fresh)
        assert np.array_equal(bits(_n(lazy.importance_grid)), bits(_n(eager.importance_grid)))
        n = lazy.correlated_update()
        assert n == eager.correlated_update()
        retraced.append(n)
        assert np.array_equal(bits(_n(lazy.photons)), bits(_n(eager.photons)))
        assert np.array_equal(_n(lazy.indices)[:n], _n(eager.indices)[:n])
        a, b = _n(lazy.light_volume), _n(eager.light_volume)   # +- atomic splats: order-dependent sums
        assert np.allclose(a, b, rtol=1e-4, atol=1e-5 * float(np.abs(a).max()))
    assert any(0 < n < lazy.n for n in retraced), retraced
    # a trace over all the samples through the stale copy re-derives it
    lazy.full_frame(); eager.full_frame()
    assert np.array_equal(bits(_n(lazy.photons)), bits(_n(eager.photons)))
    assert np.array_equal(bits(_n(lazy.light_volume)), bits(_n(eager.light_volume)))

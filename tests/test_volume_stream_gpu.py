"""cpm_volume_stream (include/cpm/cpm.h): a time-varying sequence whose steps live in HOST memory -- the upload inside the step (SURVEY 8d,
ref uniformgridcl/processors/volumesequenceplayer.cpp:94-124, dynamicvolumedifferenceanalysis.h:96-151), on the library's copy stream behind
the step before it.

  * the ring itself: every acquired volume holds its step's voxels (download) and the tracer's footprint copy of them (photons traced through
    it are the photons traced through a cpm_volume_create'd volume, bit for bit); prefetched steps are hits, absent ones upload at the
    acquire; a slot is reused only after n_slots - 1 other steps;
  * BASELINE config 5 at full size: the 32-step 256^3 sequence STREAMED from pinned host memory (prefetch t + 1, acquire t, correlated
    update) against the same sequence RESIDENT as device volumes -- after every one of the 31 transitions the re-traced indices, the
    importance grid and all photons are the same bits; the light volume is within the add-remove splat's tolerance after an update (its
    float atomics add in arrival order) and the same bits after a full frame."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LIGHT_DIR = (0.3, 0.5, -1.0)


def _n(t, dtype=None):
    a = t.detach().cpu().numpy()
    return a.view(dtype) if dtype is not None else a


@pytest.mark.parametrize("dtype,dims", [(np.uint8, (40, 36, 28)), (np.uint16, (33, 20, 17)), (np.float32, (24, 24, 24))])
def test_ring_of_slots_over_a_host_sequence(ctx, cpm, dtype, dims):
    B, S, P = cpm.binding, cpm.synthetic, cpm.pipeline
    torch = ctx.torch
    rng = np.random.default_rng(dims[0])
    shape = dims[::-1]
    if dtype == np.float32:
        steps = [rng.random(shape, dtype=np.float32) for _ in range(7)]
    else:
        steps = [rng.integers(0, np.iinfo(dtype).max, shape, dtype=dtype, endpoint=True) for _ in range(7)]
    seq = B.PinnedSequence(ctx, steps)
    vs = B.VolumeStream(ctx, steps[0], n_slots=3)
    tf = S.workspace_tf()

    def photons_through(vol):
        fr = P.PhotonFrame(ctx, vol, tf, 64, (16, 16, 16), light_travel_direction=LIGHT_DIR)
        fr.trace()
        torch.cuda.synchronize()
        return _n(fr.photons).copy()
    vs.prefetch(0, seq.steps[0])
    for t in range(7):
        if t + 1 < 7:
            vs.prefetch(t + 1, seq.steps[t + 1])
        v = vs.acquire(t)                       # resident or under way: no host voxels needed
        assert np.array_equal(v.download(), steps[t])
        if dtype != np.float32 or t < 2:
            want = photons_through(ctx.volume_create(steps[t]))
            assert np.array_equal(photons_through(v).view(np.uint32), want.view(np.uint32)), t
    info = vs.stats()
    assert (info.uploads, info.hits, info.uploads_at_acquire) == (7, 7, 0)
    assert info.bytes_uploaded == 7 * steps[0].nbytes and info.bytes_per_step == steps[0].nbytes
    # a step that is not resident uploads at the acquire (from pageable memory too); one that has been pushed out comes back the same
    v = vs.acquire(2, steps[2])
    assert np.array_equal(v.download(), steps[2])
    with pytest.raises(B.CpmError):
        vs.acquire(0)                           # long gone, and no voxels given
    a, b = vs.acquire(5), vs.acquire(6)         # still resident (the two steps before the re-upload of 2)
    assert np.array_equal(a.download(), steps[5]) and np.array_equal(b.download(), steps[6])
    info = vs.stats()
    assert (info.uploads, info.uploads_at_acquire) == (8, 1) and info.hits == 9
    torch.cuda.synchronize()
    info = vs.stats()
    assert info.uploads_timed == 8 and info.upload_ms_total > 0
    vs.close(); seq.close()


@pytest.mark.parametrize("steps", [list(range(0, 12)), list(range(11, 22)), list(range(21, 32))])   # all 31 transitions of the 32-step sequence
def test_config5_streamed_from_the_host_equals_resident(ctx, cpm, steps):
    B, S, P = cpm.binding, cpm.synthetic, cpm.pipeline
    torch = ctx.torch
    vdim, gdim, n_side, region = 256, 128, 1024, 8
    tfp = list(S.WORKSPACE_TF_POINTS)
    vols = [S.heterogeneous_volume(vdim, S.sequence_blob_center(t, 32)) for t in steps]
    resident = [ctx.volume_create(v) for v in vols]
    seq = B.PinnedSequence(ctx, vols)
    vs = B.VolumeStream(ctx, vols[0], n_slots=3)

    def mapper():
        cm = P.CorrelatedPhotonMapper(ctx, vols[0], S.tf_from_points(tfp), n_side, (gdim,) * 3, light_travel_direction=LIGHT_DIR,
                                      tf_points=tfp, incremental_threshold_percent=100.0, region=region)
        cm.full_frame()
        return cm
    a, b = mapper(), mapper()
    vs.prefetch(1, seq.steps[1])
    retraced = []
    for t in range(1, len(steps)):
        if t + 1 < len(steps):
            vs.prefetch(t + 1, seq.steps[t + 1])      # crosses PCIe while step t is computed
        b.set_volume(vs.acquire(t))
        nb = b.correlated_update()
        a.set_volume(resident[t])
        na = a.correlated_update()
        torch.cuda.synchronize()
        assert na == nb and 0 < na < a.n
        assert np.array_equal(np.sort(_n(a.indices, np.uint32)[:na]), np.sort(_n(b.indices, np.uint32)[:nb])), t
        assert np.array_equal(_n(a.photons, np.uint32), _n(b.photons, np.uint32)), t
        assert np.array_equal(_n(a.importance_grid, np.uint32), _n(b.importance_grid, np.uint32)), t
        # the light volume: both mappers update theirs with the - old / + new splat, whose float atomics add in arrival order -- two runs of
        # the SAME resident sequence differ in last bits too; held to that splat's stated tolerance here (rtol 1e-3, atol 2e-5 of the
        # maximum: DESIGN section 7) and to bit equality where the sum's order is fixed: the full frames below
        la, lb = _n(a.light_volume), _n(b.light_volume)
        assert np.allclose(la, lb, rtol=1e-3, atol=2e-5 * float(la.max())), t
        retraced.append(na)
        if t % 4 == 0 or t == len(steps) - 1:
            # a full frame of each mapper on its CURRENT volume (the streamed slot / the resident element): every photon re-traced through the
            # slot's footprint copy, brick bin + fixed-point gather -- bit for bit
            a.full_frame(); b.full_frame()
            torch.cuda.synchronize()
            assert np.array_equal(_n(a.photons, np.uint32), _n(b.photons, np.uint32)), t
            assert np.array_equal(_n(a.light_volume, np.uint32), _n(b.light_volume, np.uint32)), t
    info = vs.stats()
    assert info.uploads == len(steps) - 1 and info.uploads_at_acquire == 0 and info.hits == len(steps) - 1
    assert a.last_path == b.last_path
    vs.close(); seq.close()

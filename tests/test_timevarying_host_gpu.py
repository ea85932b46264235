"""Time-varying host layer on the GPU: the sequence processors of uniformgridcl (volume / grid players,
min/max over a sequence, difference analysis, .u3d source and export) evaluated through the C facade and
compared with the same operations issued directly on the C-ABI."""
import ctypes as C

import numpy as np
import pytest

from test_parity_gpu import _n, bits
from test_host_layer_gpu import host, host_extras  # noqa: F401  (fixtures: build and load libcpm_host after torch's HIP runtime)

pytestmark = pytest.mark.gpu

SURFACE = {  # ref uniformgridcl/processors/*.cpp constructors
    "org.inviwo.VolumeSequencePlayer": ({"volumeSequence"}, {"InterpolatedVolume"},
                                        {"time", "selectedSequenceIndex", "timePerVolume", "volumesPerSecond", "playSequence"}),
    "org.inviwo.UniformGrid3DPlayerProcessor": ({"Sequence"}, {"InterpolatedData"},
                                                {"time", "selectedSequenceIndex", "timePerElement", "frameRate", "playSequence"}),
    "org.inviwo.VolumeMinMaxCLProcessor": ({"volume", "VolumeSequenceInput"}, {"output", "UniformGrid3DVectorOut"}, {"region"}),
    "org.inviwo.DynamicVolumeDifferenceAnalysis": ({"data"}, {"DynamicDataInfo"}, {"region"}),
}
SURFACE_EXTRAS = {  # outside the workspace's path: -DCPM_HOST_EXTRAS build only
    "org.inviwo.UniformGrid3DVectorSource": (set(), {"data"}, set()),
    "org.inviwo.UniformGrid3DExport": ({"data"}, set(), set()),
    "org.inviwo.UniformGrid3DSequenceSelector": (set(), set(), set()),
}
SEQ_SIGNATURES = [("cpmh_sequence_create", C.c_void_p, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
                  ("cpmh_sequence_destroy", None, [C.c_void_p]), ("cpmh_sequence_evaluate", C.c_int, [C.c_void_p]),
                  ("cpmh_sequence_set_time_per_element", None, [C.c_void_p, C.c_float]),
                  ("cpmh_sequence_set_time", C.c_int, [C.c_void_p, C.c_float]), ("cpmh_sequence_tick", C.c_int, [C.c_void_p]),
                  ("cpmh_sequence_time", C.c_float, [C.c_void_p]), ("cpmh_sequence_max_time", C.c_float, [C.c_void_p]),
                  ("cpmh_sequence_weight", C.c_float, [C.c_void_p]),
                  ("cpmh_sequence_download", C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
                  ("cpmh_sequence_describe_surface", C.c_char_p, [C.c_void_p])]


@pytest.fixture(scope="module")
def seqlib(host):
    for name, res, args in SEQ_SIGNATURES:
        f = getattr(host, name)
        f.restype, f.argtypes = res, args
    return host


@pytest.fixture(scope="module")
def seqlib_extras(host_extras):
    for name, res, args in SEQ_SIGNATURES + [("cpmh_sequence_export", C.c_int, [C.c_void_p, C.c_int, C.c_char_p]),
                                             ("cpmh_sequence_load_grids", None, [C.c_void_p, C.c_char_p])]:
        f = getattr(host_extras, name)
        f.restype, f.argtypes = res, args
    return host_extras


def _sequence(cpm, dim=32, steps=4):
    S = cpm.synthetic
    return np.stack([S.heterogeneous_volume(dim, S.sequence_blob_center(t, steps)) for t in range(steps)])


def test_sequence_surface(seqlib, cpm):
    vols = _sequence(cpm, 16, 2)
    h = seqlib.cpmh_sequence_create(vols.ctypes.data, 0, 16, 16, 16, 2, 8)
    seen = {}
    for line in seqlib.cpmh_sequence_describe_surface(h).decode().strip().splitlines():
        cid, i, o, p = line.split("|")
        seen[cid] = (set(filter(None, i[3:].split(","))), set(filter(None, o[4:].split(","))), set(filter(None, p[5:].split(","))))
    for cid, (ins, outs, props) in SURFACE.items():
        assert cid in seen, cid
        assert ins <= seen[cid][0] and outs <= seen[cid][1] and props <= seen[cid][2], (cid, seen[cid])
    seqlib.cpmh_sequence_destroy(h)


def test_sequence_surface_of_the_extras_build(seqlib_extras, seqlib, cpm):
    vols = _sequence(cpm, 16, 2)
    h = seqlib_extras.cpmh_sequence_create(vols.ctypes.data, 0, 16, 16, 16, 2, 8)
    seen = {}
    for line in seqlib_extras.cpmh_sequence_describe_surface(h).decode().strip().splitlines():
        cid, i, o, p = line.split("|")
        seen[cid] = (set(filter(None, i[3:].split(","))), set(filter(None, o[4:].split(","))), set(filter(None, p[5:].split(","))))
    for cid, (ins, outs, props) in {**SURFACE, **SURFACE_EXTRAS}.items():
        assert cid in seen, cid
        assert ins <= seen[cid][0] and outs <= seen[cid][1] and props <= seen[cid][2], (cid, seen[cid])
    seqlib_extras.cpmh_sequence_destroy(h)
    assert not hasattr(seqlib, "cpmh_sequence_export")      # the product library: players and analyses only


def test_players_match_the_abi(seqlib, ctx, oracle, cpm, tmp_path):
    dim, steps, region = 32, 4, 8
    vols = _sequence(cpm, dim, steps)
    h = seqlib.cpmh_sequence_create(vols.ctypes.data, 0, dim, dim, dim, steps, region)
    assert h
    nb = (dim // region) ** 3
    dvols = [ctx.volume_create(v) for v in vols]
    dout = ctx.volume_create(np.zeros_like(vols[0]))
    mms, diffs = [], []
    for t in range(steps):
        mm = ctx.torch.zeros((nb, 2), dtype=ctx.torch.int16, device=ctx.device)
        ctx.volume_minmax(dvols[t], region, mm)
        df = ctx.torch.zeros(nb, dtype=ctx.torch.float32, device=ctx.device)
        ctx.volume_difference(dvols[t], dvols[(t + 1) % steps], region, df)   # next step wraps around (:62)
        mms.append(mm); diffs.append(df)
        assert np.array_equal(_n(mm, np.uint16), oracle.volume_minmax(oracle.volume(vols[t]), region))

    for time, want_index in [(0.0, 1), (0.25, 1), (1.5, 2), (2.999, 3), (3.0, 4)]:
        seqlib.cpmh_sequence_evaluate(h)  # the inport's onChange sizes the clock on first evaluation
        index = seqlib.cpmh_sequence_set_time(h, time)
        assert index == want_index
        assert seqlib.cpmh_sequence_evaluate(h) == 0
        assert seqlib.cpmh_sequence_max_time(h) == float(steps - 1)
        w = seqlib.cpmh_sequence_weight(h)
        assert abs(w - (time - int(time))) < 1e-6
        i0, i1 = index - 1, index % steps
        got = np.empty_like(vols[0])
        assert seqlib.cpmh_sequence_download(h, 0, got.ctypes.data) == 0
        ctx.volume_mix(dvols[i0], dvols[i1], w, dout)
        assert np.array_equal(got, dout.download())
        assert np.array_equal(got, oracle.volume_mix(oracle.volume(vols[i0]), oracle.volume(vols[i1]), w, vols[0]))
        gm = np.empty((nb, 2), np.uint16)
        assert seqlib.cpmh_sequence_download(h, 1, gm.ctypes.data) == 0
        om = ctx.torch.zeros((nb, 2), dtype=ctx.torch.int16, device=ctx.device)
        ctx.mix_buffers(mms[i0], mms[i1], w, om)
        assert np.array_equal(gm, _n(om, np.uint16))
        gd = np.empty(nb, np.float32)
        assert seqlib.cpmh_sequence_download(h, 2, gd.ctypes.data) == 0
        od = ctx.torch.zeros(nb, dtype=ctx.torch.float32, device=ctx.device)
        ctx.mix_buffers(diffs[i0], diffs[i1], w, od)
        assert np.array_equal(bits(gd), bits(_n(od)))

    seqlib.cpmh_sequence_destroy(h)


def test_export_and_vector_source_of_the_extras_build(seqlib_extras, ctx, cpm, tmp_path):
    """(extras build) export the analysed grids, read them back through the vector source, play them: same interpolated grid."""
    seqlib = seqlib_extras
    dim, steps, region = 32, 4, 8
    vols = _sequence(cpm, dim, steps)
    h = seqlib.cpmh_sequence_create(vols.ctypes.data, 0, dim, dim, dim, steps, region)
    nb = (dim // region) ** 3
    dvols = [ctx.volume_create(v) for v in vols]
    mms = []
    for t in range(steps):
        mm = ctx.torch.zeros((nb, 2), dtype=ctx.torch.int16, device=ctx.device)
        ctx.volume_minmax(dvols[t], region, mm)
        mms.append(mm)
    seqlib.cpmh_sequence_evaluate(h)
    path = str(tmp_path / "minmax.u3d").encode()
    assert seqlib.cpmh_sequence_export(h, 1, path) == 0
    seqlib.cpmh_sequence_load_grids(h, path)
    seqlib.cpmh_sequence_evaluate(h)
    seqlib.cpmh_sequence_set_time(h, 1.5)
    assert seqlib.cpmh_sequence_evaluate(h) == 0
    a, b = np.empty((nb, 2), np.uint16), np.empty((nb, 2), np.uint16)
    assert seqlib.cpmh_sequence_download(h, 1, a.ctypes.data) == 0
    assert seqlib.cpmh_sequence_download(h, 3, b.ctypes.data) == 0
    assert np.array_equal(a, b)
    import importlib
    seq = importlib.import_module(cpm.__name__ + ".u3d").read(path.decode())
    assert seq.data.shape == (steps, dim // region, dim // region, dim // region, 2)
    assert np.array_equal(seq.data[2].reshape(nb, 2), _n(mms[2], np.uint16))
    seqlib.cpmh_sequence_destroy(h)


def test_play_timer_wraps_around(seqlib, cpm):
    vols = _sequence(cpm, 16, 3)
    h = seqlib.cpmh_sequence_create(vols.ctypes.data, 0, 16, 16, 16, 3, 8)
    seqlib.cpmh_sequence_evaluate(h)
    seqlib.cpmh_sequence_set_time_per_element(h, 0.25)     # 3 elements -> time in [0, 0.5]
    assert abs(seqlib.cpmh_sequence_max_time(h) - 0.5) < 1e-7
    seen = []
    for _ in range(12):                                    # frame rate 10 -> +0.1 s per tick
        idx = seqlib.cpmh_sequence_tick(h)
        t = seqlib.cpmh_sequence_time(h)
        assert 0.0 <= t <= 0.5 + 1e-6
        assert idx == int(t / 0.25) % 3 + 1
        assert seqlib.cpmh_sequence_evaluate(h) == 0
        seen.append(idx)
    assert set(seen) == {1, 2, 3}
    seqlib.cpmh_sequence_destroy(h)


def test_a_sequence_kept_in_host_memory_plays_the_same_volumes(seqlib, ctx, cpm):
    """keepSequenceOnDevice = false (this build's property; the reference's elements become resident on first use): the player's two elements
    come through its ring of three device volumes, filled ahead on the library's copy stream (cpm_volume_stream) -- every displayed time,
    forth and back through the sequence and across the wrap, is the volume the resident player shows, bit for bit; every upload but the
    first frame's was started ahead of the frame that needs it."""
    dim, steps, region = 32, 6, 8
    vols = _sequence(cpm, dim, steps)
    for name, res, args in [("cpmh_sequence_keep_on_device", None, [C.c_void_p, C.c_int]), ("cpmh_sequence_stream_stats", C.c_int, [C.c_void_p, C.c_void_p])]:
        f = getattr(seqlib, name)
        f.restype, f.argtypes = res, args
    a = seqlib.cpmh_sequence_create(vols.ctypes.data, 0, dim, dim, dim, steps, region)
    b = seqlib.cpmh_sequence_create(vols.ctypes.data, 0, dim, dim, dim, steps, region)
    seqlib.cpmh_sequence_keep_on_device(b, 0)
    stats = (C.c_double * 4)()
    assert seqlib.cpmh_sequence_stream_stats(a, stats) == -1            # the resident player does not stream
    times = [0.0, 0.25, 0.5, 1.0, 1.75, 2.0, 2.5, 3.0, 3.5, 4.0, 4.9, 5.0, 4.5, 3.25, 2.0, 1.0, 0.0]
    for time in times:
        for h in (a, b):
            seqlib.cpmh_sequence_evaluate(h)
            seqlib.cpmh_sequence_set_time(h, time)
            assert seqlib.cpmh_sequence_evaluate(h) == 0
        got_a, got_b = np.empty_like(vols[0]), np.empty_like(vols[0])
        assert seqlib.cpmh_sequence_download(a, 0, got_a.ctypes.data) == 0 and seqlib.cpmh_sequence_download(b, 0, got_b.ctypes.data) == 0
        assert np.array_equal(got_a, got_b), time
        for kind, dt, shape in ((1, np.uint16, ((dim // region) ** 3, 2)), (2, np.float32, ((dim // region) ** 3,))):
            ga, gb = np.empty(shape, dt), np.empty(shape, dt)
            assert seqlib.cpmh_sequence_download(a, kind, ga.ctypes.data) == 0 and seqlib.cpmh_sequence_download(b, kind, gb.ctypes.data) == 0
            assert np.array_equal(ga.view(np.uint8), gb.view(np.uint8)), (time, kind)
    ctx.torch.cuda.synchronize()
    assert seqlib.cpmh_sequence_stream_stats(b, stats) == 0
    uploads, upload_ms, bytes_per_step, late = int(stats[0]), float(stats[1]), int(stats[2]), int(stats[3])
    assert bytes_per_step == dim ** 3 and uploads >= steps and upload_ms > 0
    assert late <= 4          # the very first frame's two elements, and the turn-around (the walk back is not what the player prefetches for)
    seqlib.cpmh_sequence_destroy(a)
    seqlib.cpmh_sequence_destroy(b)

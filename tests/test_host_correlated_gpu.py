"""The C++ processors' importance branch without a host round trip (ProgressivePhotonTracerCL: fusedImportanceBranch;
PhotonToLightVolumeProcessorCL: replaced records instead of the whole-buffer snapshot), the time-varying form of the
workspace (sequence players -> importance -> tracer -> light volume) and the timing harness bench.py uses."""
import ctypes as C

import numpy as np
import pytest

from test_parity_gpu import _n, bits
from test_host_layer_gpu import host, Net, _light  # noqa: F401
from test_timevarying_host_gpu import seqlib, _sequence  # noqa: F401

pytestmark = pytest.mark.gpu

BASE = [(0.0, 1, 1, 1, 0.0), (0.45, 1, 0.5, 0.2, 0.0), (0.55, 0.6, 0.3, 0.1, 0.05), (0.8, 0.9, 0.2, 0.3, 0.4), (1.0, 0.1, 0.6, 0.7, 0.5)]
EDIT = list(BASE)
EDIT[3] = (0.85,) + BASE[3][1:]


@pytest.fixture(scope="module")
def hostx(seqlib):
    lib = seqlib
    for name, res, args in [("cpmh_bench_tf_edits", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
                            ("cpmh_bench_full_frames", C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
                            ("cpmh_attach_sequence", C.c_int, [C.c_void_p, C.c_void_p]),
                            ("cpmh_last_tracer_decision", C.c_char_p, [C.c_void_p]),
                            ("cpmh_path_costs", None, [C.c_void_p, C.c_void_p]),
                            ("cpmh_sequence_step", C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p])]:
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    return lib


@pytest.mark.parametrize("max_scattering", [1, 2])
def test_fused_branch_equals_launch_by_launch(hostx, cpm, max_scattering):
    """Two networks, one with fusedImportanceBranch off (count read on the host between select and trace, snapshot copy kept):
    same photons, same number re-traced, light volumes within the add-remove tolerance, over three edits."""
    host = hostx
    S = cpm.synthetic
    vol = S.heterogeneous_volume(64)
    pos, d = _light(cpm, (0.3, 0.5, -1.0))
    a = Net(host, vol, 128, pos, d, BASE, correlated=True, max_scattering=max_scattering)
    b = Net(host, vol, 128, pos, d, BASE, correlated=True, max_scattering=max_scattering)
    assert host.cpmh_set_property_float(b.h, b"tracer", b"fusedImportanceBranch", 0.0) == 0
    assert host.cpmh_set_property_string(a.h, b"tracer", b"importanceBranchPolicy", b"always") == 0   # the branch itself is under test
    # c: the device-count branch as separate launches (selection, compaction, cpm_trace_selected) instead of the one-launch form
    c = Net(host, vol, 128, pos, d, BASE, correlated=True, max_scattering=max_scattering)
    assert host.cpmh_set_property_string(c.h, b"tracer", b"importanceBranchPolicy", b"always") == 0
    assert host.cpmh_set_property_float(c.h, b"tracer", b"retraceInImportancePass", 0.0) == 0
    c.evaluate(first=True)
    assert host.cpmh_set_property_float(c.h, b"lightvolume", b"incrementalRecomputationThreshold", 100.0) == 0
    for net in (a, b):
        net.evaluate(first=True)
        assert host.cpmh_set_property_float(net.h, b"lightvolume", b"incrementalRecomputationThreshold", 100.0) == 0
    # the legacy network needs one evaluation to take its first snapshot: a full frame follows its first edit
    for k, pts in enumerate((EDIT, BASE, EDIT)):
        for net in (a, b, c):
            net.set_tf(pts)
            net.evaluate()
        na, nb = host.cpmh_n_recomputed(a.h), host.cpmh_n_recomputed(b.h)
        assert na == nb == host.cpmh_n_recomputed(c.h) > 0
        assert host.cpmh_last_light_volume_path(c.h) == b"incremental"
        assert np.array_equal(bits(a.photons()), bits(c.photons()))
        lc, _, _ = c.light_volume()
        assert host.cpmh_last_light_volume_path(a.h) == b"incremental"
        assert np.array_equal(bits(a.photons()), bits(b.photons()))
        fresh = Net(host, vol, 128, pos, d, pts, correlated=False, max_scattering=max_scattering)
        fresh.evaluate(first=True)
        assert np.array_equal(bits(a.photons()), bits(fresh.photons()))
        lv_full, _, _ = fresh.light_volume()
        la, _, _ = a.light_volume()
        lb, _, _ = b.light_volume()
        np.testing.assert_allclose(la, lv_full, rtol=1e-3, atol=2e-5 * float(lv_full.max()))
        np.testing.assert_allclose(lb, lv_full, rtol=1e-3, atol=2e-5 * float(lv_full.max()))
        np.testing.assert_allclose(lc, lv_full, rtol=1e-3, atol=2e-5 * float(lv_full.max()))
        fresh.close()
    assert host.cpmh_last_light_volume_path(b.h) == b"incremental"
    # an edit that changes nothing re-traces nothing and leaves the volume alone
    a.set_tf(EDIT)
    a.evaluate()
    assert host.cpmh_n_recomputed(a.h) == 0 and host.cpmh_last_light_volume_path(a.h) == b"unchanged"
    la2, _, _ = a.light_volume()
    assert np.array_equal(bits(la2), bits(la))
    # above the add-remove threshold the light volume is rebuilt
    assert host.cpmh_set_property_float(a.h, b"lightvolume", b"incrementalRecomputationThreshold", 0.01) == 0
    a.set_tf(BASE)
    a.evaluate()
    assert host.cpmh_n_recomputed(a.h) == na and host.cpmh_last_light_volume_path(a.h) == b"full"
    a.close(); b.close(); c.close()


def test_timer_continuation_updates_the_light_volume_itself(hostx, cpm):
    """a correlated update continued on the refinement timer (only the Progressive flag set) is an add-remove on the
    light volume itself, not an estimate to be averaged in.  Two networks run the same budget-limited update, one with
    progressiveAccumulation off: the continuation must not depend on that property, and the photons end on a from-scratch
    network's.  (The radius schedule advances on every timer tick, as in the reference -- tracercl.cpp:252-260 -- so the
    volume itself is not comparable with a single-radius frame.)"""
    host = hostx
    S = cpm.synthetic
    vol = S.heterogeneous_volume(64)
    pos, d = _light(cpm, (0.3, 0.5, -1.0))
    nets = [Net(host, vol, 128, pos, d, BASE, correlated=True) for _ in range(2)]
    assert host.cpmh_set_property_float(nets[1].h, b"lightvolume", b"progressiveAccumulation", 0.0) == 0
    results = []
    for net in nets:
        assert host.cpmh_set_property_float(net.h, b"tracer", b"maxIncrementalPhotonsToUpdate", 10.0) == 0
        net.evaluate(first=True)
        assert host.cpmh_set_property_float(net.h, b"lightvolume", b"incrementalRecomputationThreshold", 100.0) == 0
        lv0, _, _ = net.light_volume()
        net.set_tf(EDIT)
        net.evaluate()
        rounds, paths = 1, [host.cpmh_last_light_volume_path(net.h)]
        while host.cpmh_remaining(net.h) > 0:
            assert host.cpmh_refine(net.h) >= 1      # onTimerEvent + process: the path that sets only the Progressive flag
            paths.append(host.cpmh_last_light_volume_path(net.h))
            rounds += 1
        assert rounds > 2 and b"progressive" not in paths and paths[-1] == b"incremental", paths
        lv, _, _ = net.light_volume()
        assert np.abs(lv - lv0).max() > 1e-3 * float(lv0.max())   # the batches did reach the volume
        results.append((net.photons(), lv))
    fresh = Net(host, vol, 128, pos, d, EDIT, correlated=False)
    fresh.evaluate(first=True)
    assert np.array_equal(bits(results[0][0]), bits(fresh.photons()))
    assert np.array_equal(bits(results[0][0]), bits(results[1][0]))
    np.testing.assert_allclose(results[0][1], results[1][1], rtol=1e-3, atol=4e-5 * float(results[1][1].max()))
    for net in nets + [fresh]:
        net.close()


def test_time_varying_network(hostx, cpm):
    """The workspace's time-varying form: players feed the interpolated volume, min/max and difference grids; per displayed
    time the importance processor's time-varying branch (prev / new min-max, difference) drives a correlated re-trace that
    lands on the photons of a from-scratch network of that time's volume -- up to the photons whose path only grazes changed
    bricks (bricks without apron).  Times inside one sequence interval: there the difference grid the reference
    wires in (|v_(t+1) - v_t|, uniformgridcl/processors/dynamicvolumedifferenceanalysis.h:96-151) is the change on display."""
    host = hostx
    S = cpm.synthetic
    dim, steps, region = 64, 4, 8
    vols = _sequence(cpm, dim, steps)
    seq = host.cpmh_sequence_create(vols.ctypes.data, 0, dim, dim, dim, steps, region)
    assert seq
    pos, d = _light(cpm, (0.3, 0.5, -1.0))
    net = Net(host, vols[0], 128, pos, d, S.WORKSPACE_TF_POINTS, correlated=True)
    assert host.cpmh_attach_sequence(net.h, seq) == 0
    assert host.cpmh_set_property_string(net.h, b"tracer", b"importanceBranchPolicy", b"always") == 0
    net.evaluate(first=True)
    assert host.cpmh_set_property_float(net.h, b"lightvolume", b"incrementalRecomputationThreshold", 100.0) == 0
    n_total = host.cpmh_n_photons(net.h)
    times = (C.c_double * 2)()
    for t in (0.3, 0.6, 0.9):
        n = host.cpmh_sequence_step(net.h, seq, float(t), C.byref(times))
        assert 0 < n < n_total, (t, n)
        assert times[0] > 0 and times[1] > 0
        assert host.cpmh_last_light_volume_path(net.h) == b"incremental"
        shown = np.zeros_like(vols[0])
        assert host.cpmh_sequence_download(seq, 0, shown.ctypes.data) == 0
        assert not np.array_equal(shown, vols[0]) and not np.array_equal(shown, vols[1])
        fresh = Net(host, shown, 128, pos, d, S.WORKSPACE_TF_POINTS, correlated=False)
        fresh.evaluate(first=True)
        got, want = net.photons(), fresh.photons()
        stale = (bits(got) != bits(want)).any(axis=1).mean()
        assert stale < 0.02, (t, stale)
        lv, _, _ = net.light_volume()
        lv_full, _, _ = fresh.light_volume()
        assert np.abs(lv - lv_full).max() <= 0.05 * float(lv_full.max())
        fresh.close()
    net.close()
    host.cpmh_sequence_destroy(seq)


def test_timing_harness(hostx, cpm):
    host = hostx
    S = cpm.synthetic
    vol = S.heterogeneous_volume(64)
    pos, d = _light(cpm, (0.3, 0.5, -1.0))
    net = Net(host, vol, 128, pos, d, BASE, correlated=True)
    assert host.cpmh_set_property_string(net.h, b"tracer", b"importanceBranchPolicy", b"always") == 0
    net.evaluate(first=True)
    a = np.ascontiguousarray(np.asarray(EDIT, np.float32))
    b = np.ascontiguousarray(np.asarray(BASE, np.float32))
    reps = 6
    ms = (C.c_double * reps)()
    n = (C.c_int * reps)()
    assert host.cpmh_bench_tf_edits(net.h, a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], reps, C.byref(ms), C.byref(n)) == 0
    assert all(m > 0 for m in ms) and len(set(n)) == 1 and n[0] > 0
    assert host.cpmh_bench_full_frames(net.h, reps, C.byref(ms)) == 0
    assert all(m > 0 for m in ms) and host.cpmh_n_recomputed(net.h) == -1
    net.close()


def test_adaptive_branch_takes_the_measured_cheaper_path(hostx, cpm):
    """importanceBranchPolicy = adaptive (the default): the first TF edit goes through the importance branch (nothing measured yet); from
    then on the tracer compares the GPU-timeline cost of branch + add-remove with that of a full frame and serves the edit
    with the cheaper one.  Whatever it takes, the photons are those of a from-scratch network."""
    host = hostx
    S = cpm.synthetic
    vol = S.heterogeneous_volume(64)
    pos, d = _light(cpm, (0.3, 0.5, -1.0))
    net = Net(host, vol, 128, pos, d, BASE, correlated=True)
    net.evaluate(first=True)
    assert host.cpmh_last_tracer_decision(net.h) == b"full frame"
    assert host.cpmh_set_property_float(net.h, b"lightvolume", b"incrementalRecomputationThreshold", 100.0) == 0
    costs = (C.c_float * 4)()
    decisions = []
    net.evaluate(first=True)                              # a second full frame: the first sample of a path is dropped
    for k in range(6):
        pts = EDIT if k % 2 == 0 else BASE
        host.cpmh_path_costs(net.h, C.byref(costs))
        before = list(costs)
        net.set_tf(pts)
        net.evaluate()                                    # (ends with a device synchronisation: the spans are complete)
        dec = host.cpmh_last_tracer_decision(net.h)
        decisions.append(dec)
        known = all(c >= 0 for c in before)
        if not known:
            assert dec == b"importance branch", (k, before)
        else:
            want_full = before[2] + before[3] > before[0] + before[1]
            assert dec.startswith(b"full frame (measured") == want_full, (k, before, dec)
        n = host.cpmh_n_recomputed(net.h)
        assert (n == -1) == dec.startswith(b"full frame")
        fresh = Net(host, vol, 128, pos, d, pts, correlated=False)
        fresh.evaluate(first=True)
        assert np.array_equal(bits(net.photons()), bits(fresh.photons()))
        lv, _, _ = net.light_volume()
        lv_full, _, _ = fresh.light_volume()
        np.testing.assert_allclose(lv, lv_full, rtol=1e-3, atol=2e-5 * float(lv_full.max()))
        fresh.close()
    assert decisions[0] == decisions[1] == b"importance branch"       # the first sample of a path is a warm-up, not a measurement
    host.cpmh_path_costs(net.h, C.byref(costs))
    assert all(c > 0 for c in costs)
    net.close()


def test_failed_importance_launch_falls_back_to_a_full_frame(hostx, cpm):
    """A select / retrace call that fails after appending its tiles (injected: cpm_debug_fail_next_select) must not leave tiles no
    kernel wrote in the selection: the finish publishes a count of 0 and reports the failure, the tracer serves the edit with a
    full frame -- the photons of a from-scratch evaluation -- and the next edit takes the branch again."""
    host = hostx
    host.cpmh_debug_fail_next_select.argtypes = [C.c_void_p]
    host.cpmh_debug_fail_next_select.restype = None
    S = cpm.synthetic
    vol = S.heterogeneous_volume(64)
    pos, d = _light(cpm, (0.3, 0.5, -1.0))
    for one_launch in (1.0, 0.0):
        net = Net(host, vol, 128, pos, d, BASE, correlated=True)
        assert host.cpmh_set_property_string(net.h, b"tracer", b"importanceBranchPolicy", b"always") == 0
        assert host.cpmh_set_property_float(net.h, b"tracer", b"retraceInImportancePass", one_launch) == 0
        net.evaluate(first=True)
        assert host.cpmh_set_property_float(net.h, b"lightvolume", b"incrementalRecomputationThreshold", 100.0) == 0
        host.cpmh_debug_fail_next_select(net.h)
        net.set_tf(EDIT)
        net.evaluate()
        assert host.cpmh_last_tracer_decision(net.h) == b"full frame (the importance branch failed)"
        assert host.cpmh_n_recomputed(net.h) == -1 and host.cpmh_last_light_volume_path(net.h) == b"full"
        fresh = Net(host, vol, 128, pos, d, EDIT, correlated=False)
        fresh.evaluate(first=True)
        assert np.array_equal(bits(net.photons()), bits(fresh.photons()))
        a, _, _ = net.light_volume()
        b, _, _ = fresh.light_volume()
        assert np.array_equal(bits(a), bits(b))
        net.set_tf(BASE)                                   # the branch works again
        net.evaluate()
        assert host.cpmh_last_tracer_decision(net.h) == b"importance branch" and host.cpmh_n_recomputed(net.h) > 0
        net.close(); fresh.close()

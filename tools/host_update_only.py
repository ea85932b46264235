#!/usr/bin/env python3
"""The correlated update through the C++ processors, alone, for rocprofv3: config3 = 60 TF edits (alternating edit / revert) of
the 256^3 / 1 M photon network; config5 = the time-varying network stepped a quarter of a sequence step at a time.
usage: tools/host_update_only.py [config3|config5] [reps] [volume dim] [lattice side]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd, importlib
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
what = sys.argv[1] if len(sys.argv) > 1 else "config3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
vdim = int(sys.argv[3]) if len(sys.argv) > 3 else 256
nside = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
ctx = B.Context(0)
H = importlib.import_module(cpm_amd.__name__ + ".hostlayer")
hl = H.load()
vol = S.heterogeneous_volume(vdim)
base = list(S.WORKSPACE_TF_POINTS); edit = list(base); edit[3] = (0.26,) + base[3][1:]
d = P._normalize((0.3, 0.5, -1.0)); pos = np.array([0.5] * 3, np.float32) - np.float32(2.0) * d
if what == "config3":
    net = H.HostNetwork(hl, vol, nside, pos, d, base, size_option=2, correlated=True)
    import os
    if os.environ.get("CPM_RETRACE_IN_PASS") == "0":
        net.set_float("tracer", "retraceInImportancePass", 0.0)
    net.evaluate(first=True)
    full = net.bench_full_frames(reps)
    net.set_string("tracer", "importanceBranchPolicy", "never")
    never, _ = net.bench_tf_edits(edit, base, reps)
    net.set_string("tracer", "importanceBranchPolicy", "adaptive")
    auto, na = net.bench_tf_edits(edit, base, reps)
    print(f"edits served by full frames only {np.median(never[5:]):.4f} ms; adaptive {np.median(auto[5:]):.4f} ms ({np.mean(na[5:] < 0):.0%} full frames), costs {net.path_costs()}")
    net.set_string("tracer", "importanceBranchPolicy", "always")
    ms, n = net.bench_tf_edits(edit, base, reps)
    print(f"[{vdim}^3, {nside}^2 photons] ", end="")
    print(f"config3: full frame {np.median(full[5:]):.4f} ms, update {np.median(ms[5:]):.4f} ms (p10 {np.percentile(ms[5:], 10):.4f}, p90 {np.percentile(ms[5:], 90):.4f}), "
          f"re-traced {n[-1]} of {net.n_photons} = {n[-1] / net.n_photons:.4%}, path {net.last_path}")
else:
    n_steps = 8
    seq_np = np.stack([S.heterogeneous_volume(256, S.sequence_blob_center(t, 32)) for t in range(n_steps)])
    seq = H.HostSequence(hl, seq_np)
    net = H.HostNetwork(hl, seq_np[0], 1024, pos, d, base, size_option=2, correlated=True)
    seq.attach(net)
    net.evaluate(first=True)
    rows = [seq.step(net, 0.25 * (k % (4 * (n_steps - 1)))) for k in range(1, reps)]
    rows = rows[3:]
    print(f"config5: players {np.median([r[1] for r in rows]):.4f} ms, update {np.median([r[2] for r in rows]):.4f} ms, "
          f"re-traced {np.mean([r[0] for r in rows]) / net.n_photons:.4%}, path {net.last_path}")

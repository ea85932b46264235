#!/usr/bin/env python3
"""Where a config-3 update's host time goes: the host clock since the TF edit after each processor has returned (medians)."""
import sys
sys.path.insert(0, '.')
import numpy as np, cpm_amd, importlib
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = B.Context(0)
H = importlib.import_module(cpm_amd.__name__ + ".hostlayer")
hl = H.load()
vol = S.heterogeneous_volume(256)
base = list(S.WORKSPACE_TF_POINTS); edit = list(base); edit[3] = (0.26,) + base[3][1:]
d = P._normalize((0.3, 0.5, -1.0)); pos = np.array([0.5] * 3, np.float32) - np.float32(2.0) * d
net = H.HostNetwork(hl, vol, 1024, pos, d, base, size_option=2, correlated=True)
net.evaluate(first=True)
net.set_string("tracer", "importanceBranchPolicy", "always")
net.bench_tf_edits(edit, base, 10)
tl = net.bench_tf_edits_timeline(edit, base, reps)[5:]
m = np.median(tl, axis=0) * 1e3
print("us since the edit (median): property set %.1f | importance.process %.1f | tracer.process %.1f | lightVolume.process %.1f | idle %.1f" % tuple(m))

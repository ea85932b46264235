"""The workspace point's TF-edit update (importance pass + re-trace of both lights, delta splat) through the C++ processors: ms per update and
the kernels' shares.  usage: [CPM_HOST_PHOTON_LAYOUT=interleaved] python tools/ws_tf_edit.py [edits]"""
import importlib
import sys
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
S, P = cpm_amd.synthetic, cpm_amd.pipeline
edits = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.zeros(1, device="cuda")
H = importlib.import_module(cpm_amd.__name__ + ".hostlayer")
hl = H.load()
lights = []
for w in ((-90.045471, 104.828, 312.07489), (94.269867, 148.44716, 302.45557)):
    d = P._normalize(tuple(-x for x in w))
    lights.append((np.array([0.5] * 3, np.float32) - np.float32(2.0) * d, d))
base = list(S.WORKSPACE_TF_POINTS)
edit = list(base)
edit[3] = (0.26,) + base[3][1:]
net = H.HostNetwork(hl, S.heterogeneous_volume((512, 512, 96)), 1024, lights[0][0], lights[0][1], base, size_option=2, correlated=True)
net.add_light(*lights[1])
net.set_clip(73, 512, 7, 512, 0, 96)
net.evaluate(first=True)
net.bench_frames_back_to_back(10)
ms, n = net.bench_tf_edits(edit, base, edits)
print(f"tf edit at the workspace point: median {float(np.median(ms[10:])):.4f} ms  (fraction re-traced {float(np.mean(np.maximum(n[10:], 0))) / (2 * 1024 * 1024):.5f}, served by {net.last_decision})")
if hasattr(net, "profile_tf_edits"):
    prof = net.profile_tf_edits(edit, base, 20)
    print("  " + "  ".join(f"{k.split('(')[0][-44:]} {v * 1e3:.1f}" for k, v in sorted(prof.items(), key=lambda kv: -kv[1])[:8]))

"""Per-kernel times of the workspace's operating point (C++ processors, library HIP-event hook).  usage: tools/ws_kernels.py [frames]
(with LD_PRELOAD=build/variants/<name>.so for a variant build: tools/ws_variants.sh)"""
import importlib
import json
import sys
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
S, P = cpm_amd.synthetic, cpm_amd.pipeline
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.zeros(1, device="cuda")
H = importlib.import_module(cpm_amd.__name__ + ".hostlayer")
hl = H.load()
lights = []
for w in ((-90.045471, 104.828, 312.07489), (94.269867, 148.44716, 302.45557)):
    d = P._normalize(tuple(-x for x in w))
    lights.append((np.array([0.5] * 3, np.float32) - np.float32(2.0) * d, d))
net = H.HostNetwork(hl, S.heterogeneous_volume((512, 512, 96)), 1024, lights[0][0], lights[0][1], list(S.WORKSPACE_TF_POINTS), size_option=2, correlated=True)
net.add_light(*lights[1])
net.set_clip(73, 512, 7, 512, 0, 96)
net.evaluate(first=True)
ms, _ = net.bench_frames_back_to_back(frames)
prof = net.profile_full_frames(30)
print(f"workspace point: {ms:.4f} ms per frame; " + "  ".join(f"{k.split('(')[0][-40:]} {v * 1e3:.1f}" for k, v in sorted(prof.items(), key=lambda kv: -kv[1])[:5]))

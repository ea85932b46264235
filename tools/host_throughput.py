"""The C++ Processor/Port network's frame: throughput (frames back to back, one synchronisation) and latency (from an idle
device) at config 2 and at the workspace's own operating point (two lights, 512 x 512 x 96), next to the Python driver's frame.
  python tools/host_throughput.py [reps]"""
import importlib
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import cpm_amd  # noqa: E402

S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
hostlayer = importlib.import_module(cpm_amd.__name__ + ".hostlayer")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.zeros(1, device="cuda")
ctx = B.Context(0)
hl = hostlayer.load()
LIGHT_DIR = (0.3, 0.5, -1.0)
d = P._normalize(LIGHT_DIR)
lpos = np.array([0.5, 0.5, 0.5], np.float32) - np.float32(2.0) * d


def report(name, net, n_photons):
    net.evaluate(first=True)
    net.bench_frames_back_to_back(20)
    thr, host = net.bench_frames_back_to_back(reps)
    lat = float(np.median(net.bench_full_frames(60)[10:]))
    print(f"{name}: throughput {thr:.4f} ms/frame ({n_photons / thr / 1e3:.0f} Mphotons/s), host enqueue {host:.4f} ms/frame, latency from idle {lat:.4f} ms")
    print("    kernels (us per frame):", {k: round(v * 1e3, 1) for k, v in sorted(net.profile_full_frames(50).items(), key=lambda kv: -kv[1])})


vol = S.heterogeneous_volume(256)
fr = P.PhotonFrame(ctx, vol, S.workspace_tf(), 1024, (128,) * 3, light_travel_direction=LIGHT_DIR)
for _ in range(20):
    fr.frame_fast()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    fr.frame_fast()
torch.cuda.synchronize()
print(f"python driver config2: {(time.perf_counter() - t) / reps * 1e3:.4f} ms/frame")
for corr in (False, True):
    net = hostlayer.HostNetwork(hl, vol, 1024, lpos, d, list(S.WORKSPACE_TF_POINTS), size_option=2, correlated=corr)
    report(f"host network config2 (correlated={corr})", net, 1 << 20)
    net.close()
del fr
# the workspace point
import os
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
LW = [(-90.045471, 104.828, 312.07489), (94.269867, 148.44716, 302.45557)]
lights = []
for w in LW:
    dd = P._normalize(tuple(-x for x in w))
    lights.append((np.array([0.5, 0.5, 0.5], np.float32) - np.float32(2.0) * dd, dd))
volw = S.heterogeneous_volume((512, 512, 96))
for form in ("fast", "gather", "splat"):
    net = hostlayer.HostNetwork(hl, volw, 1024, lights[0][0], lights[0][1], list(S.WORKSPACE_TF_POINTS), size_option=2, correlated=True)
    net.add_light(*lights[1])
    net.set_clip(73, 512, 7, 512, 0, 96)
    net.set_string("lightvolume", "formulation", form)
    report(f"workspace point (2 x 1024^2, 512x512x96, lv 256x256x48, formulation={form})", net, 2 << 20)
    net.close()

"""Repeated launches of one stage on config 2's photons (for rocprofv3 --pmc passes).
usage: python3 tools/stage_only.py trace|bin_fast|gather_fast|frame_fast|frame [repeats] [workload]"""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
stage = sys.argv[1] if len(sys.argv) > 1 else "trace"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wl = sys.argv[3] if len(sys.argv) > 3 else "config2"
vdim, nside, gdim = {"config2": (256, 1024, 128), "config4": (512, 2048, 256), "config1": (64, 256, 32)}[wl]
import os
ctx = B.Context(0)
tf_alpha = os.environ.get("CPM_TF_ALPHA")  # constant-alpha transfer function instead of the workspace's (1.0 = one Woodcock step per photon)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(vdim), (S.homogeneous_tf(float(tf_alpha)) if tf_alpha else S.workspace_tf()), nside, (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
fr.frame(); fr.frame_fast()
if stage == "frame_fast" and os.environ.get("CPM_RECORDS", "planar") == "planar":   # the bench's default: two-plane records
    fr.set_planar_records(True)
    fr.frame_fast()
torch.cuda.synchronize()
fn = {"trace": fr.trace, "bin_fast": fr.bin_fast, "gather_fast": fr.gather_fast, "frame_fast": fr.frame_fast, "frame": fr.frame,
      "bin": fr.bin, "gather": fr.gather}[stage]
for _ in range(reps):
    fn()
torch.cuda.synchronize()

import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), S.workspace_tf(), 1024, (128,)*3, light_travel_direction=(0.3, 0.5, -1.0))
for _ in range(3):
    fr.frame()
torch.cuda.synchronize()

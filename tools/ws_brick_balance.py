"""How evenly the brick gather's work items fall on its workgroups at the workspace's operating point (one light standing in for two:
2 x 1024^2 photons through the 512 x 512 x 96 volume, 256 x 256 x 48 light volume, boxes of 6 x 6 x 2 candidates).
Reads the brick table the bin wrote (records per brick: since the halo form a photon is filed once, under the brick of its box's low corner)
and replays the launch's static hand-out -- item i to workgroup i mod G -- and alternatives.  usage (GPU box): python tools/ws_brick_balance.py"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume((512, 512, 96)), S.workspace_tf(), (1024, 2048), (256, 256, 48), light_travel_direction=(0.28, -0.32, -0.95))
fr.set_planar_records(True)
fr.trace(); fr.bin_fast(); fr.gather_fast()
torch.cuda.synchronize()
t = fr.brick_table.cpu().numpy().view(np.uint32)
# table: [0, nb] starts | 4 meta | items; nb from the layout: 16 x 16 x 8 bricks -> 16 * 16 * 6
nb = 16 * 16 * 6
starts = t[:nb + 1].astype(np.int64)
counts = np.diff(starts)
n_items = int(t[nb + 1 + 2])
items = t[nb + 5:nb + 5 + n_items]
c = counts[items]
print(f"records {starts[-1]}  bricks {nb}  non-empty {n_items}  records per non-empty brick: mean {c.mean():.0f} median {np.median(c):.0f} max {c.max()}  p90 {np.percentile(c, 90):.0f}")
G = 512
cost = 3.0 + c / 2048.0 * 2.2      # a brick: ~3 us of fixed latency chain + ~2.2 us per pass of 2048 records (rough)
for name, order in (("brick order (the launch's)", np.arange(n_items)), ("largest first", np.argsort(-c))):
    load = np.zeros(G)
    for k, i in enumerate(order):
        load[k % G] += cost[i]
    lpt = np.zeros(G)
    for i in np.argsort(-c):
        lpt[np.argmin(lpt)] += cost[i]
    print(f"{name:28s} static i mod {G}: max {load.max():.1f} mean {load.mean():.1f} us (model)   greedy largest-first: max {lpt.max():.1f}")
for _ in range(10):
    fr.gather_fast()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    fr.gather_fast()
e1.record(); torch.cuda.synchronize()
print(f"gather alone: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
hist = np.bincount(np.minimum(c // 2048, 12))
print("bricks by passes of 2048 records:", hist.tolist())

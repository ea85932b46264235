"""Per-kernel times (the library's HIP-event hook) of one rank's frame at config 4's size under a given sharding: the dense form
(cpm_gather_fast_marked) and the segment form (cpm_gather_fast_segment).  usage: python tools/shard_kernels.py [world] [rank] [tiles|range]"""
import importlib
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
sh = importlib.import_module(cpm_amd.__name__ + ".sharding")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = sys.argv[3] if len(sys.argv) > 3 else "range"
vdim, (nx, ny), gdim = 512, (2048, 2048), 256
ctx = B.Context(0)
vol = ctx.volume_create(S.heterogeneous_volume(vdim))
tf = S.workspace_tf()
n_total = nx * ny
if kind == "tiles":
    shard = sh.shard_tiles(n_total, rank, world)
else:
    lo, hi = sh.shard_range(n_total, rank, world)
    shard = np.arange(lo, hi, dtype=np.int64)
fr = P.PhotonFrame(ctx, vol, tf, (nx, ny), (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0), photon_indices=shard)
fr.set_planar_records(True)
nb = ((gdim + 3) // 4) ** 3
room = (nb + 63) & ~63
marks = torch.zeros(nb + 16, dtype=torch.uint8, device=ctx.device)
buf = torch.empty(16 + room * 272, dtype=torch.uint8, device=ctx.device)
ctl = torch.zeros(2, dtype=torch.int32, device=ctx.device)
mail = torch.zeros(1, dtype=torch.int64, device=ctx.device)
seg = B.BricklistSegment(buf.data_ptr(), 8192, room, 7, 1, ctl.data_ptr(), mail.data_ptr())
for name, gather in (("dense", lambda: fr.gather_fast(nonzero_bricks=marks)), ("segment", lambda: fr.gather_fast_segment(seg))):
    for _ in range(10):
        fr.trace(); fr.bin_fast(); gather()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(50):
        fr.trace(); fr.bin_fast(); gather()
    k = ctx.profile_collect(); ctx.profile_enable(False)
    print(name, {kk.split("(")[0][:40]: round(tot / calls * 1e3, 1) for kk, (tot, calls) in k.items()}, "sum", round(sum(tot for tot, _ in k.values()) / 50 * 1e3, 1), flush=True)
tbl = fr.brick_table.cpu().numpy().view(np.uint32)
print("photons", fr.n, "records", int((fr.photons[:, 0] < 1e30).sum().item()) if False else "-")
# the bricks' loads: table[0 .. nb] = brick starts
nbk = None
for cand in (8192, 4096, 2048):
    if tbl[cand] == tbl[:cand + 1].max() and tbl[cand] > 0 and (np.diff(tbl[:cand + 1].astype(np.int64)) >= 0).all():
        nbk = cand
        break
if nbk:
    cnt = np.diff(tbl[:nbk + 1].astype(np.int64))
    items = np.nonzero(cnt)[0]
    c = cnt[items]
    print("gather bricks", nbk, "non-empty", items.size, "records", int(c.sum()), "max", int(c.max()), "mean", round(float(c.mean()), 1),
          "p50/p90/p99", [int(np.percentile(c, q)) for q in (50, 90, 99)], "bricks > 2048 records:", int((c > 2048).sum()), "> 4096:", int((c > 4096).sum()))
    heavy_pos = np.nonzero(c > 2048)[0]
    print("heavy items' positions in the list (of", items.size, "):", heavy_pos[:40].tolist())
    # per workgroup (512, round-robin): records in all
    wg = np.zeros(512, np.int64)
    np.add.at(wg, np.arange(items.size) % 512, c)
    print("records per workgroup: max", int(wg.max()), "mean", round(float(wg.mean()), 1), "items of the max WG:", c[np.arange(items.size) % 512 == int(wg.argmax())].tolist())

"""Experiment: the brick gather with its work items (non-empty bricks) reordered by record count.
usage: python tools/gather_lpt_exp.py [config2|config4]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
vdim, nside, gdim = {"config2": (256, 1024, 128), "config4": (512, 2048, 256)}[wl]
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(vdim), S.workspace_tf(), nside, (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
fr.frame_fast(); torch.cuda.synchronize()
want = fr.light_volume.clone()
t = fr.brick_table
nb = (t.numel() - 5) // 2
n_items = int(t[nb + 1 + 2].item())
items = t[nb + 5: nb + 5 + n_items].clone().long()
counts = (t[items + 1] - t[items]).long()
c = counts.cpu().numpy()
print(f"{wl}: {nb} bricks, {n_items} with records; records per brick: median {np.median(c):.0f}, p90 {np.percentile(c, 90):.0f}, max {c.max()}, total {c.sum()}")
G = 2 * 256
def loads(order):
    w = np.zeros(G)
    for i, it in enumerate(order): w[i % G] += np.ceil(c[it] / 2048) * 2048 + 1500   # batches of 2048 records + a fixed cost per brick
    return w
def timeit(fn, reps=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def run(order, name):
    t[nb + 5: nb + 5 + n_items] = items[torch.as_tensor(order)].to(t.dtype)
    us = timeit(fr.gather_fast)
    torch.cuda.synchronize()
    ok = torch.equal(fr.light_volume.view(torch.int32), want.view(torch.int32))
    w = loads(order)
    print(f"  {name:50s} gather {us:6.1f} us   (modelled max / mean load {w.max() / w.mean():.2f}; volume identical: {ok})")
run(np.arange(n_items), "brick order (as built)")
desc = np.argsort(-c, kind="stable")
run(desc, "records descending, round-robin")
snake = np.empty(n_items, dtype=np.int64)
rounds = [desc[i:i + G] for i in range(0, n_items, G)]
snake = np.concatenate([r if k % 2 == 0 else r[::-1] for k, r in enumerate(rounds)])
run(snake, "records descending, snake")
bucket = np.argsort(-np.floor(np.log2(np.maximum(c, 1))), kind="stable")
run(bucket, "log2(records) buckets descending")
run(np.argsort(c, kind="stable"), "records ascending")
ident = np.arange(n_items)
rounds = [ident[i:i + G] for i in range(0, n_items, G)]
run(np.concatenate([r if k % 2 == 0 else r[::-1] for k, r in enumerate(rounds)]), "brick order, snake (odd rounds reversed)")
rng = np.random.default_rng(0)
run(rng.permutation(n_items), "random permutation")
# rotation: round k shifted by k * 37 workgroups
run(np.concatenate([np.roll(r, 37 * k) for k, r in enumerate(rounds)]), "brick order, round k rotated by 37 k")

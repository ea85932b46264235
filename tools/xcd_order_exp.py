"""Which chunks an XCD traces (review item 7): the product's table deals the 4096-sample lattice tiles round-robin to the 8 XCDs
(every XCD touches the whole volume); here each XCD gets a CONTIGUOUS range of the lattice instead -- equal counts, or equal measured
cost (prefix sums of the chunk costs) -- its heaviest chunks first either way.  The table is written through the measurement hook;
the trace alone is timed back to back, photons compared with the default order's.
usage: python tools/xcd_order_exp.py [config2|config4]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
vdim, nside, gdim = {"config2": (256, 1024, 128), "config4": (512, 2048, 256)}[wl]
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(vdim), S.workspace_tf(), nside, (gdim,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
fr.trace()                       # measured launch + the product's re-sort
torch.cuda.synchronize()
ref = fr.photons.clone()
order0, _, _ = fr.trace_order.read()
# the chunk costs: measure once more without letting the update clear them
ctx.trace_set_order(fr.trace_order, True)
ctx.trace(fr.vol, fr.tf, fr.aabb, fr.params, fr.light_samples, fr.isect, fr.rng, fr.photons)
ctx.trace_set_order(None)
_, cost, launches = fr.trace_order.read()
n = order0.size
assert launches >= 1 and cost.sum() > 0
cost = cost.astype(np.float64)


def time_table(table, label):
    fr.trace_order.write(table)
    fr._traces_since_order = 1   # (no re-measure, no re-sort)
    for _ in range(5):
        fr.trace()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fr.trace()
    e1.record(); torch.cuda.synchronize()
    same = bool(torch.equal(fr.photons.view(torch.int32), ref.view(torch.int32)))
    print(f"{label:64s} {e0.elapsed_time(e1) / 200 * 1e3:7.2f} us   photons identical: {same}")


def heavy_first(chunks):
    """an XCD's list: its heaviest eighth first, the rest in lattice order (the product's rule)"""
    chunks = np.asarray(chunks)
    k = max(1, chunks.size // 8)
    heavy = chunks[np.argsort(-cost[chunks], kind="stable")[:k]]
    hs = set(heavy.tolist())
    return np.concatenate([np.sort(heavy), np.array([c for c in chunks if c not in hs], np.int64)])


def interleave(lists):
    """position p of XCD x -> workgroup 8 p + x; XCDs that run out take what the longest lists still hold (their tails)"""
    per = n // 8
    lists = [list(l) for l in lists]
    spill = []
    for l in lists:
        if len(l) > per:
            spill.extend(l[per:]); del l[per:]
    for l in lists:
        while len(l) < per and spill:
            l.append(spill.pop())
    table = np.empty(n, np.int64)
    for x in range(8):
        table[x::8][: len(lists[x])] = lists[x]
    return table


time_table(order0, "product: tiles dealt round-robin to the XCDs, heavy first")
ranges = np.array_split(np.arange(n), 8)
time_table(interleave([heavy_first(r) for r in ranges]), "contiguous eighths of the lattice (equal counts), heavy first")
cs = np.cumsum(cost)
cuts = [0] + [int(np.searchsorted(cs, cs[-1] * k / 8)) for k in range(1, 8)] + [n]
time_table(interleave([heavy_first(np.arange(cuts[k], cuts[k + 1])) for k in range(8)]), "contiguous ranges of equal measured cost, heavy first")
# sixteenths dealt to XCD x = k mod 8: two separate slabs per XCD (half-way between the two)
six = np.array_split(np.arange(n), 16)
time_table(interleave([heavy_first(np.concatenate([six[x], six[x + 8]])) for x in range(8)]), "two contiguous sixteenths per XCD, heavy first")
time_table(order0, "product again")

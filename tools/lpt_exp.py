"""Trace launch times under cpm_trace_order: default order, measured order, with the table read back.
usage: python tools/lpt_exp.py"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), S.workspace_tf(), 1024, (128,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
fr.adaptive_order = False
def timeit(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(f"no order object: trace {timeit(fr.trace):.1f} us")
order = ctx.trace_order_create(fr.n)
ctx.trace_set_order(order)
print(f"order object, default table (costs collected): trace {timeit(fr.trace):.1f} us")
t, c, l = order.read()
print(f"  launches {l}, cost/launch: min {c.min() / l:.1f} median {np.median(c) / l:.1f} p90 {np.percentile(c, 90) / l:.1f} max {c.max() / l:.1f}")
order.update()
t2, _, _ = order.read()
print(f"  table changed in {(t != t2).sum()} of {len(t)} places; first 16 of XCD 0: {t2[0:128:8].tolist()}")
print(f"measured order (still measuring): trace {timeit(fr.trace):.1f} us")
ctx.trace_set_order(order, False)
print(f"measured order (not measuring): trace {timeit(fr.trace):.1f} us")
ctx.profile_reset(); ctx.profile_enable(True)
ctx.trace_set_order(order, True); fr.trace(); order.update(); fr.trace(); order.update()
k = ctx.profile_collect(); ctx.profile_enable(False)
for kn, (tot, calls) in k.items():
    print(f"    {kn:40s} {tot / calls * 1e3:8.1f} us x {calls}")
ctx.trace_set_order(order, False)
print(f"measured order (second update, not measuring): trace {timeit(fr.trace):.1f} us")
ctx.trace_set_order(None)
print(f"no order object: trace {timeit(fr.trace):.1f} us")

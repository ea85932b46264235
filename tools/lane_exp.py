"""Lanes sorted by last launch's steps inside every chunk (cpm_trace_order): wave-iterations and launch time, on the workspace TF
and on constant-alpha TFs (long walks).  usage: python tools/lane_exp.py"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, tf in (("workspace TF", S.workspace_tf()), ("alpha 0.05", S.homogeneous_tf(0.05)), ("alpha 0.01", S.homogeneous_tf(0.01))):
    fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), tf, 1024, (128,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
    fr.adaptive_order = False
    base = timeit(fr.trace)
    order = ctx.trace_order_create(fr.n)
    out = []
    for rnd in range(2):
        ctx.trace_set_order(order, True); fr.trace(); torch.cuda.synchronize()
        _, c, l = order.read()
        order.update()
        ctx.trace_set_order(order, False)
        out.append((int(c.sum()), timeit(fr.trace)))
    ctx.trace_set_order(None)
    order.close()
    print(f"{name}: no order {base:.1f} us | lattice lanes: {out[0][0]} wave-iterations | sorted lanes + chunk order: {out[1][0]} wave-iterations, {out[1][1]:.1f} us")

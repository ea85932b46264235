"""Experiment: the one-launch importance + re-trace kernel with its tiles ordered by measured cost."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
ctx.lib.cpm_debug_set_retrace_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
ctx.lib.cpm_debug_set_retrace_order.restype = None
vol = S.heterogeneous_volume(256)
base = list(S.WORKSPACE_TF_POINTS); edit = list(base); edit[3] = (0.26,) + base[3][1:]
cm = P.CorrelatedPhotonMapper(ctx, vol, S.workspace_tf(), 1024, (128,) * 3, light_travel_direction=(0.3, 0.5, -1.0), tf_points=base)
cm.full_frame()
reps = 40
def run(n, prof):
    ctx.profile_reset(); ctx.profile_enable(prof)
    for rep in range(n):
        cm.set_transfer_function(edit if rep % 2 == 0 else base)
        k = cm.correlated_update()
    torch.cuda.synchronize()
    kern = ctx.profile_collect() if prof else {}
    ctx.profile_enable(False)
    return kern
def kernel_us(label):
    run(6, False)
    kern = run(reps, True)
    for name, (ms, calls) in kern.items():
        if "importance_retrace" in name:
            print(f"{label:40s} importance_retrace_kernel {ms / calls * 1e3:7.2f} us")
kernel_us("tile = blockIdx")
ntiles = 1048576 // int(__import__("os").environ.get("TILE", "512"))
cost = torch.zeros(ntiles, dtype=torch.int32, device="cuda")
ctx.lib.cpm_debug_set_retrace_order(ctx.h, None, cost.data_ptr())
run(8, False)
ctx.lib.cpm_debug_set_retrace_order(ctx.h, None, None)
c = cost.cpu().numpy().astype(np.int64)
print(f"tile cost (100 MHz ticks, max over waves and launches): min {c.min()} median {int(np.median(c))} p90 {int(np.percentile(c, 90))} max {c.max()}")
def top(frac):
    k = int(ntiles * frac)
    srt = np.argsort(-c, kind="stable")
    heavy = np.sort(srt[:k]); rest = np.sort(srt[k:])
    return np.concatenate([heavy, rest])
def top_xcd(frac):
    out = np.empty(ntiles, dtype=np.int64)
    for x in range(8):
        mine = np.arange(x, ntiles, 8)
        k = int(len(mine) * frac)
        srt = mine[np.argsort(-c[mine], kind="stable")]
        out[x::8] = np.concatenate([np.sort(srt[:k]), np.sort(srt[k:])])
    return out
for name, order in (("heaviest first", np.argsort(-c, kind="stable")), ("heaviest 1/8 first, rest in order", top(0.125)), ("heaviest 1/4 first, rest in order", top(0.25)),
                    ("heaviest 1/2 first, rest in order", top(0.5)), ("per XCD: heaviest 1/8 first", top_xcd(0.125)), ("per XCD: heaviest 1/4 first", top_xcd(0.25))):
    o = torch.as_tensor(order.astype(np.int32)).cuda()
    ctx.lib.cpm_debug_set_retrace_order(ctx.h, o.data_ptr(), None)
    kernel_us(name)
    ctx.lib.cpm_debug_set_retrace_order(ctx.h, None, None)

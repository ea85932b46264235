"""Trace kernel time against the number of Woodcock steps: config 2's rays with transfer functions of constant alpha
(1.0 -> one step per photon: prologue + epilogue only; smaller alpha -> more steps).  usage: python tools/trace_exp.py"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
vol = S.heterogeneous_volume(256)
def run(tf, label):
    fr = P.PhotonFrame(ctx, vol, tf, 1024, (128,) * 3, light_travel_direction=(0.3, 0.5, -1.0))
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    ctx.set_step_counter(counter); fr.trace(); torch.cuda.synchronize(); ctx.set_step_counter(None)
    steps = int(counter.item())
    for _ in range(5): fr.trace()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fr.trace()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{label:28s} steps/photon {steps / fr.n:7.2f}  trace {us:8.1f} us  {steps / us / 1e3:7.2f} Gsteps/s")
run(S.homogeneous_tf(1.0), "alpha 1.0")
run(S.homogeneous_tf(0.5), "alpha 0.5")
run(S.workspace_tf(), "workspace TF (config 2)")
run(S.homogeneous_tf(0.1), "alpha 0.1")
run(S.homogeneous_tf(0.02), "alpha 0.02")
run(S.homogeneous_tf(0.0), "alpha 0 (all exit)")

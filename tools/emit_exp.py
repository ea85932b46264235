"""Trace with the samples read from buffers against the samples evaluated in the kernel (cpm_trace_emitted), under the launch order."""
import sys
sys.path.insert(0, '.')
import torch, cpm_amd
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
def timeit(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
ph = {}
for emit in (False, True, False, True):
    fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), S.workspace_tf(), 1024, (128,) * 3, light_travel_direction=(0.3, 0.5, -1.0), emit_in_tracer=emit)
    t = timeit(fr.trace); f = timeit(fr.frame_fast)
    ph[emit] = fr.photons.clone()
    print(f"emit_in_tracer={emit}: trace {t:.1f} us, frame {f:.1f} us")
print("photons identical:", bool(torch.equal(ph[False].view(torch.int32), ph[True].view(torch.int32))))

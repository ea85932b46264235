import sys; sys.path.insert(0,'.')
import numpy as np, torch, cpm_amd
B=cpm_amd.binding; S=cpm_amd.synthetic
ctx=B.Context(0)
for dt,n in ((np.uint8,256),(np.uint16,256),(np.float32,256),(np.uint8,512)):
    v=(np.random.default_rng(0).integers(0,255,(n,n,n))).astype(dt)
    h=ctx.volume_create(v)
    d=torch.from_numpy(v.view(np.uint8) if dt!=np.uint8 else v).to('cuda')
    for how in ('device update',):
        for _ in range(3): h.update(d)
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): h.update(d)
        e1.record(); torch.cuda.synchronize()
        us=e0.elapsed_time(e1)/20*1e3
        print(dt.__name__, n, how, f"{us:.1f} us  {v.nbytes*6/us/1e6:.2f} TB/s (6 x volume bytes)")

import sys
sys.path.insert(0, '.')
import numpy as np, torch, cpm_amd, ctypes
S, P, B = cpm_amd.synthetic, cpm_amd.pipeline, cpm_amd.binding
ctx = B.Context(0)
fr = P.PhotonFrame(ctx, S.heterogeneous_volume(256), S.workspace_tf(), 1024, (128,)*3, light_travel_direction=(0.3, 0.5, -1.0))
for _ in range(3): fr.frame()
torch.cuda.synchronize()
nb = 32*32*32
st = torch.zeros((nb, 4), dtype=torch.int64, device='cuda')
ctx.lib.cpm_debug_set_gather_stamps(ctx.h, ctypes.c_void_p(st.data_ptr()))
fr.gather(); torch.cuda.synchronize()
ctx.lib.cpm_debug_set_gather_stamps(ctx.h, None)
a = st.cpu().numpy()
t0 = a[:,0].min(); start = (a[:,0]-t0)/100.0; end = (a[:,1]-t0)/100.0; dur = end-start; rec = a[:,2] & 0xffffffff; mx = a[:,2] >> 32; xcc = a[:,3]
print('kernel span us', end.max(), ' waves', nb)
print('dur us percentiles 50/90/99/99.9/max', np.percentile(dur,[50,90,99,99.9,100]))
heavy = rec > 0
print('nonempty bricks', heavy.sum(), 'records: mean', rec[heavy].mean(), 'max', rec.max())
print('dur of nonempty: mean', dur[heavy].mean(), 'max', dur[heavy].max(), ' corr(dur,rec)', np.corrcoef(dur[heavy], rec[heavy])[0,1])
for q in (0, 100, 500, 1000, 2000, 4000):
    m = (rec >= q) & (rec < (q*2 if q else 100))
    if m.any(): print(f'  rec in [{q},{q*2 if q else 100}): n={m.sum():6d} dur mean {dur[m].mean():8.1f} us  max {dur[m].max():8.1f}')
print('max-lane tests: mean over heavy', mx[heavy].mean(), 'max', mx.max(), ' sum tests', rec.sum(), ' sum of max*64', (mx*64).sum())
print('us per max-lane-test (heavy):', (dur[heavy]/np.maximum(mx[heavy],1)).mean())
print('per XCC: waves, sum dur (us), last end (us)')
for k in range(8):
    m = xcc == k
    print('  xcc', k, m.sum(), round(dur[m].sum()), round(end[m].max(),1), 'heavy', (heavy & m).sum())
# timeline: number of waves running at time t
ts = np.linspace(0, end.max(), 21)
print('resident waves over time:', [int(((start <= t) & (end > t)).sum()) for t in ts])
idx = np.argsort(-dur)[:8]
print('slowest bricks (gb, bx,by,bz, rec, dur, start):', [(int(i), int(i%32), int((i//32)%32), int(i//1024), int(rec[i]), round(float(dur[i]),1), round(float(start[i]),1)) for i in idx])

import sys, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from oracle_binding import Oracle
o=Oracle()
rng=np.random.default_rng(0)
n=1<<20
ph=np.zeros((n,8),np.float32); ph[:,:3]=rng.random((n,3),dtype=np.float32); ph[::9,:3]=3.402823466e+38
g=o.grid((128,)*3,1)
for T in (1,8,16,64,256,256,16):
    o.set_threads(T)
    o.bin(ph,n,g)
    t=time.perf_counter(); r=o.bin(ph,n,g); print(T,'threads bin', round(time.perf_counter()-t,3),'s', flush=True)

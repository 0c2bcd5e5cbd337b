import sys, time, io, contextlib
import numpy as np
sys.path.insert(0, '.')
from impdar_amd import _hip
from impdar_amd.lib.NoInitRadarData import NoInitRadarDataFiltering
_hip.load(); ctx=_hip.context(); lib=_hip.load()
snum,tnum=4096,10000
rng=np.random.default_rng(0)
x=rng.standard_normal((snum,tnum)).astype(np.float32)
dist=np.hstack(([0.],np.cumsum(0.6+0.8*rng.random(tnum-1))))/1000.
for rep in range(3):
    d=NoInitRadarDataFiltering(); d.data,(d.snum,d.tnum)=x.copy(),x.shape; d.dt,d.dist=1e-8,dist.copy()
    d.travel_time=np.arange(snum)*1e-2
    for a in ['lat','long','x_coord','y_coord','decday','pressure','elev']: setattr(d,a,np.arange(tnum,dtype=float))
    d.trig=np.zeros(tnum)
    T=[time.perf_counter()]
    def mark(): lib.impdar_ctx_sync(ctx); T.append(time.perf_counter())
    with contextlib.redirect_stdout(io.StringIO()):
        d.to_device(); mark()
        d.vertical_band_pass(2.,10.); mark()
        d.constant_space(1.0); mark()
        d.migrate('stolt',htaper=100,vtaper=1000); mark()
        d.from_device(); mark()
    print(' '.join('%s %.1f' % (n,(b-a)*1e3) for n,a,b in zip(['up','vbp','cspace','stolt','down'],T[:-1],T[1:])), 'total %.1f ms' % ((T[-1]-T[0])*1e3))

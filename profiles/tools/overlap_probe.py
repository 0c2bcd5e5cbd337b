"""Does the producer stream really overlap the diffraction sum?  Times an 8-rank interior block (1152 output
traces, 1256-trace input shard) alone, with the prep serial, and pipelined.  NR=<ranks> selects the plan's rank
count (plans for > 1 rank use the LDS-free prep).  python profiles/tools/overlap_probe.py"""
import sys, time, numpy as np
sys.path.insert(0,'.')
from impdar_amd import _hip, parallel, synth
from impdar_amd.kirchhoff import KirchhoffPlan
snum,tnum,vel=4096,10000,1.69e8
geo=synth.geometry(snum,tnum); tt=geo['travel_time']/1e6
_hip.load(); ctx=_hip.context(0)
data=np.random.default_rng(0).standard_normal((snum,tnum)).astype(np.float32)
import os
plan=KirchhoffPlan(ctx,np.float32,snum,tnum,geo['dist'],geo['travel_time'],vel,False,'fast',nranks=int(os.environ.get('NR','1')))
d_full=_hip.DeviceArray.from_host(ctx,data)
xlo,xhi=5000,6152; jlo,jhi=5024,6280
d_in=_hip.DeviceArray.from_host(ctx,np.ascontiguousarray(data[:,jlo:jhi]))
d_out=_hip.DeviceArray(ctx,(snum,xhi-xlo),np.float32)
plan.prep(d_full,tnum,0,tnum); plan.migrate(d_out,xlo,xhi); plan.sync()
def timeit(fn,K=60):
    for _ in range(5): fn()
    plan.sync(); t0=time.perf_counter()
    for _ in range(K): fn()
    plan.sync(); return (time.perf_counter()-t0)/K*1e3
print('migrate only, back to back      %.3f ms'%timeit(lambda: plan.migrate(d_out,xlo,xhi)))
def f2(): plan.migrate(d_out,xlo,xhi); plan.sync()
print('migrate + sync each             %.3f ms'%timeit(f2))
def f3(): plan.prep(d_in,jhi-jlo,jlo,jhi-jlo); plan.migrate(d_out,xlo,xhi)
print('prep(shard)+migrate pipelined   %.3f ms'%timeit(f3))
def f4(): plan.prep(d_in,jhi-jlo,jlo,jhi-jlo); plan.migrate(d_out,xlo,xhi); plan.sync()
print('prep(shard)+migrate + sync each %.3f ms'%timeit(f4))
def f5(): plan.prep(d_full,tnum,0,tnum); plan.migrate(d_out,xlo,xhi); plan.sync()
print('prep(full)+migrate + sync each  %.3f ms (kernel %.3f)'%(timeit(f5), plan.last_ms()[2]))
def f6(): plan.prep(d_in,1,jlo,0); plan.migrate(d_out,xlo,xhi)
print('prep(nothing)+migrate pipelined %.3f ms'%timeit(f6))
import os

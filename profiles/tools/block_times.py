import sys, numpy as np, time
sys.path.insert(0,'.')
from impdar_amd import _hip, parallel, synth
from impdar_amd.kirchhoff import KirchhoffPlan
snum,tnum,vel=4096,10000,1.69e8
geo=synth.geometry(snum,tnum); tt=geo['travel_time']/1e6
lib=_hip.load(); ctx=_hip.context(0)
data=np.random.default_rng(0).standard_normal((snum,tnum)).astype(np.float32)
plan=KirchhoffPlan(ctx,np.float32,snum,tnum,geo['dist'],geo['travel_time'],vel,False,'fast',nranks=1)
d_in=_hip.DeviceArray.from_host(ctx,data); d_out=_hip.DeviceArray(ctx,(snum,tnum),np.float32)
def run(xlo,xhi,reps=5):
    ts=[]
    for _ in range(reps):
        plan.prep(d_in,tnum,0,tnum); plan.migrate(d_out,xlo,xhi); plan.sync(); ts.append(plan.last_ms()[2])
    return min(ts)
full=run(0,tnum)
print('full',full)
for n in (2,4,8):
    _,_,blocks,pairs=parallel.plan_blocks(tt,1.0,vel,tnum,n)
    ts=[run(lo,hi) for lo,hi in blocks]
    print(n,'blocks',[b[1]-b[0] for b in blocks],'ms',[round(t,3) for t in ts],'max',max(ts),'ideal',full/n,'eff',full/n/max(ts))

for ws in ([2889,2112,2112,2887],[2696,2304,2304,2696],[2792,2208,2208,2792]):
    e=np.concatenate([[0],np.cumsum(ws)]); blocks=[(int(e[i]),int(e[i+1])) for i in range(len(ws))]
    ts=[run(lo,hi) for lo,hi in blocks]
    print('custom',ws,'ms',[round(t,3) for t in ts],'max',round(max(ts),3),'eff',round(full/len(ws)/max(ts),3))

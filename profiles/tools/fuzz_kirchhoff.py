"""Random-geometry sweep of the fast Kirchhoff kernels against the C oracle (GPU box).
    python profiles/tools/fuzz_kirchhoff.py <seed> <cases>"""
import sys, numpy as np
sys.path.insert(0,'.')
from impdar_amd import synth, _hip
from impdar_amd.kirchhoff import migrate_resident
from oracle import c_oracle
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
ctx=_hip.context()
bad=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 40):
    snum=int(rng.integers(2,1500)); tnum=int(rng.integers(1,700))
    dt=float(10**rng.uniform(-9.3,-7.5)); dx=float(10**rng.uniform(-1.3,0.9)); vel=float(rng.uniform(0.6e8,3e8))
    t0=float(rng.choice([0.0, dt*1e6, -3*dt*1e6, 17.3*dt*1e6]))
    near=bool(rng.integers(0,2))
    geo=synth.geometry(snum,tnum,dt=dt,dx=dx,t0_us=t0)
    data=rng.standard_normal((snum,tnum)).astype(np.float32)
    sa=2*dx/(vel*dt)
    try:
        out,mode,_=migrate_resident(ctx,data,geo['dist'],geo['travel_time'],vel,nearfield=near,mode='fast')
    except NotImplementedError as e:
        print(it,'unsupported sa=%.2f'%sa); continue
    want=c_oracle.kirchhoff(data,geo['travel_time'],geo['dist'],vel,near)
    nrm=np.linalg.norm(want)
    err=np.linalg.norm(out-want)/nrm if nrm>0 else np.abs(out).max()
    flag='' if err<1e-4 else '  <<<<<< BAD'
    if flag: bad+=1
    print(it,snum,tnum,'dt %.2e dx %.2f v %.2e t0 %.3g near %d sa %.2f err %.2e'%(dt,dx,vel,t0,near,sa,err),flag)
print('bad',bad)

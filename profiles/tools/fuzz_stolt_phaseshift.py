"""Random-geometry sweep of the Stolt and phase-shift paths against the NumPy oracle (GPU box).
    python profiles/tools/fuzz_stolt_phaseshift.py <seed> <cases>"""
import sys, numpy as np, io, contextlib
sys.path.insert(0,'.')
from impdar_amd import synth
from impdar_amd.lib.RadarData import RadarData
from oracle import mig_oracle
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
def mk(x,geo):
    d=RadarData(None); d.data,(d.snum,d.tnum)=x.copy(),x.shape
    d.travel_time,d.dist,d.trace_int,d.dt=geo['travel_time'],geo['dist'],geo['trace_int'],geo['dt']; return d
bad=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 30):
    snum=int(rng.integers(4,700)); tnum=int(rng.integers(2,500))
    dt=float(10**rng.uniform(-9,-7.6)); dx=float(10**rng.uniform(-1,0.8)); vel=float(rng.uniform(0.8e8,2.8e8))
    ht=int(rng.integers(1,40)); vt=int(rng.integers(1,60))
    dtype=rng.choice([np.float32,np.float64])
    geo=synth.geometry(snum,tnum,dt=dt,dx=dx)
    x=rng.standard_normal((snum,tnum)).astype(dtype)
    # stolt
    d=mk(x,geo)
    with contextlib.redirect_stdout(io.StringIO()): d.migrate('stolt',vel=vel,htaper=ht,vtaper=vt)
    want=mig_oracle.stolt(x,dt,geo['trace_int'],geo['dist'],vel,ht,vt)
    e1=np.linalg.norm(d.data-want)/max(np.linalg.norm(want),1e-300); ok1=(d.data.dtype==want.dtype) and d.data.shape==want.shape
    # phase shift: const or table
    if rng.integers(0,2):
        v=vel
    else:
        Rp=2.8e8*geo['travel_time'][-1]*1e-6/2.
        v=np.array([[vel,0.],[vel,0.3*Rp],[min(vel*1.2,2.9e8),0.6*Rp],[min(vel*1.3,2.9e8),1.3*Rp]])
    d=mk(x,geo)
    try:
        with contextlib.redirect_stdout(io.StringIO()): d.migrate('phsh',vel=v,htaper=ht,vtaper=vt)
        want=mig_oracle.phase_shift(x,dt,geo['trace_int'],geo['travel_time'],geo['dist'],v,ht,vt)
        e2=np.linalg.norm(d.data-want)/max(np.linalg.norm(want),1e-300)
    except ValueError as e:
        e2=-1
    tol1=1e-4 if dtype==np.float32 else 1e-9; tol2=2e-4 if dtype==np.float32 else 1e-8
    flag='' if (e1<tol1 and e2<tol2 and ok1) else '  <<<<< BAD'
    if flag: bad+=1
    print(it,snum,tnum,np.dtype(dtype).name,'dt %.1e dx %.2f ht %d vt %d'%(dt,dx,ht,vt),'table' if hasattr(v,'__len__') else 'const','stolt %.1e phsh %.1e'%(e1,e2),flag)
print('bad',bad)

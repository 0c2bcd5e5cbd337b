"""Workgroup-residency diagnostic for kirch_quad_kernel (not part of the product).

Build a stamped library and run on a GPU box:
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -DKQ_STAMP \\
          -c impdar_amd/csrc/kirchhoff.hip -o /tmp/k.o
    hipcc --offload-arch=gfx950 -shared -fPIC impdar_amd/csrc/{api,comm,stolt,phaseshift}.o /tmp/k.o \\
          -o build/libimpdar_stamp.so -L/opt/rocm/lib -lrocfft -lrccl
    IMPDAR_HIP_LIB=$PWD/build/libimpdar_stamp.so python profiles/tools/stamp_run.py
Every workgroup records s_memrealtime at entry/exit, its XCC id and its step count; the script
prints the number of resident workgroups over time, microseconds per step per sample chunk and
the per-XCD finish times (r01 result: 750-768 of 768 slots occupied until the last 8 % of the
kernel, 0.26 us per 8x... step, XCDs finish within +-3 %).
"""
import ctypes as C, os, sys, json
import numpy as np
sys.path.insert(0, os.getcwd())
from impdar_amd import _hip, synth
from impdar_amd.kirchhoff import KirchhoffPlan
lib=_hip.load(); ctx=_hip.context(0)
snum,tnum=4096,10000
geo=synth.geometry(snum,tnum)
x=np.random.default_rng(0).standard_normal((snum,tnum)).astype(np.float32)
plan=KirchhoffPlan(ctx,np.float32,snum,tnum,geo['dist'],geo['travel_time'],mode='fast')
d_in=_hip.DeviceArray.from_host(ctx,x); d_out=_hip.DeviceArray(ctx,(snum,tnum),np.float32)
for _ in range(3):
    plan.prep(d_in,tnum,0,tnum); plan.migrate(d_out,0,tnum); plan.sync()
n=(1<<22)//8
buf=np.zeros(n,dtype=np.uint64)
lib.impdar_kirch_debug_stamps.argtypes=[C.c_void_p,C.c_void_p,C.c_size_t]
rc=lib.impdar_kirch_debug_stamps(plan.h, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
print('rc',rc, plan.last_ms())
st=buf.reshape(-1,4); st=st[st[:,0]>0]
t0=st[:,0].min(); start=(st[:,0]-t0)/100.0; end=(st[:,1]-t0)/100.0   # 100 MHz -> us
xcc=(st[:,2]>>32).astype(int); hwid=(st[:,2]&0xffffffff).astype(int)
chunk=(st[:,3]>>32).astype(int); steps=(st[:,3]&0xffffffff).astype(int)
dur=end-start
print('WGs',len(st),'kernel span us',end.max(),'mean dur',dur.mean(),'max',dur.max())
print('us per step by chunk:',[round(float(np.median(dur[chunk==c]/steps[chunk==c])),4) for c in range(16)])
print('median dur by chunk:',[round(float(np.median(dur[chunk==c])),1) for c in range(16)])
# occupancy timeline: number of resident WGs over time
T=np.linspace(0,end.max(),60)
occ=[int(((start<=t)&(end>t)).sum()) for t in T]
print('resident WGs over time:',occ)
for x in range(8):
    m=xcc==x
    print('xcc',x,'WGs',int(m.sum()),'last end',round(float(end[m].max()),1),'sum dur',round(float(dur[m].sum()),0))
cu=(hwid>>8)&0xf; se=(hwid>>13)&0x7
slots = 256 * (2 if os.environ.get('IMPDAR_KIRCH_NH', '1') == '1' else 1)
print('slot utilisation: sum of workgroup durations / (%d slots x kernel span) = %.4f' % (slots, dur.sum() / (slots * end.max())))
T2 = np.linspace(0.80 * end.max(), end.max(), 21)
print('resident WGs over the last 20 %:', [int(((start <= t) & (end > t)).sum()) for t in T2])

import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import synth, _hip
from impdar_amd.kirchhoff import KirchhoffPlan, migrate_resident
snum, tnum = 4096, 10000
geo = synth.geometry(snum, tnum)
ctx = _hip.context()
for dtype in (np.float32, np.float64):
    x = np.random.default_rng(0).standard_normal((snum, tnum)).astype(dtype)
    p = KirchhoffPlan(ctx, dtype, snum, tnum, geo['dist'], geo['travel_time'], mode='auto')
    print(np.dtype(dtype).name, p.mode, p.kernel, flush=True)
    p.destroy()
    for i in range(3):
        t0 = time.perf_counter()
        out, mode, ms = migrate_resident(ctx, x, geo['dist'], geo['travel_time'], mode='auto')
        print('  resident wall %.1f ms, events %s' % ((time.perf_counter() - t0) * 1e3, ms), flush=True)

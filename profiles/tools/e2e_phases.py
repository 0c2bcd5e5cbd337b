import os, sys, time, io, contextlib
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import synth
from impdar_amd.lib.RadarData import RadarData
snum, tnum = 4096, 10000
geo = synth.geometry(snum, tnum)
for dtype in (np.float32, np.float64):
    x = np.random.default_rng(0).standard_normal((snum, tnum)).astype(dtype)
    for i in range(4):
        d = RadarData(None); d.data, d.snum, d.tnum = x, snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            d.migrate('kirch')
        print(np.dtype(dtype).name, 'wall %.1f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)

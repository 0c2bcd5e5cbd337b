"""RadarData.migrate('kirch') on a float32 host radargram at config 3, 4 calls (for a rocprofv3 timeline of the one-shot call)."""
import os, sys, time, io, contextlib
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import synth
from impdar_amd.lib.RadarData import RadarData
snum, tnum = 4096, 10000
geo = synth.geometry(snum, tnum)
if os.environ.get('E2E_DATA', 'synthetic') == 'synthetic':      # bench.py's radargram (diffractors: mostly smooth -- the chip clocks higher on it than on noise)
    x = synth.diffractor_radargram(snum, tnum, vel=1.69e8, dtype=np.float32, trace_lo=0, trace_hi=tnum, chunk=128, threads=8)
else:
    x = np.random.default_rng(0).standard_normal((snum, tnum)).astype(np.float32)
for i in range(int(os.environ.get("E2E_CALLS", "4"))):
    d = RadarData(None); d.data, d.snum, d.tnum = x, snum, tnum
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        d.migrate('kirch', vel=float(os.environ.get('E2E_VEL', '1.69e8')))
    print('float32 wall %.1f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
    time.sleep(0.05)

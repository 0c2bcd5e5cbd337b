"""One-shot phase-shift / Stolt calls on host arrays with IMPDAR_TRACE milestones: where the wall time of `impproc migrate` goes."""
import os, sys, time, io, contextlib
sys.path.insert(0, os.getcwd())
os.environ['IMPDAR_TRACE'] = '1'
import numpy as np
from impdar_amd import synth
from impdar_amd.lib.RadarData import RadarData
for kind, n in (('phsh', 8192), ('stolt', 4096)):
    geo = synth.geometry(n, n)
    x = np.random.default_rng(0).standard_normal((n, n)).astype(np.float32)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    tab = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
    for i in range(4):
        d = RadarData(None); d.data, d.snum, d.tnum = x, n, n
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        sys.stderr.write('---- %s call %d\n' % (kind, i)); sys.stderr.flush()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            if kind == 'phsh':
                d.migrate('phsh', vel=tab, htaper=100, vtaper=1000)
            else:
                d.migrate('stolt', vel=1.68e8, htaper=100, vtaper=1000)
        sys.stderr.write('---- wall %.1f ms\n' % ((time.perf_counter() - t0) * 1e3)); sys.stderr.flush()

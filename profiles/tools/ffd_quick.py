"""Wall time of the v(x,z) Fourier finite-difference phase shift, chain-in-one-workgroup against launch-per-step
(IMPDAR_FFD_CHAIN=0), on a 3-column velocity table.  usage: ffd_quick.py [snum] [tnum]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from impdar_amd import synth
from impdar_amd.lib.RadarData import RadarData
from impdar_amd.lib import migrationlib

snum = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tnum = int(sys.argv[2]) if len(sys.argv) > 2 else 512
geo = synth.geometry(snum, tnum, dx=5.0)
rng = np.random.default_rng(0)
data = rng.standard_normal((snum, tnum))
Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
xs = np.linspace(0., geo['dist'][-1] * 1e3, 5)
rows = []
for x in xs:
    for v, z in ((1.69e8, 0.), (1.72e8 + 2e5 * x / xs[-1], 0.6 * Rp), (1.8e8, 1.3 * Rp)):
        rows.append((v, z, x))
vel = np.array(rows)
res = {}
for chain in ('1', '0'):
    os.environ['IMPDAR_FFD_CHAIN'] = chain
    best = None
    for rep in range(2):
        d = RadarData(None)
        d.data, d.snum, d.tnum = data.copy(), snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            migrationlib.migrationPhaseShift(d, vel=vel, htaper=10, vtaper=10)
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    res[chain] = (best, d.data)
nt = 1 << int(np.ceil(np.log2(snum)))
steps = snum * nt
a, b = res['1'][1], res['0'][1]
print('snum %d tnum %d nt %d: %d steps; one workgroup %.3f s (%.2f us/step), launch per step %.3f s (%.2f us/step); '
      'finite %s, max rel difference %.3g'
      % (snum, tnum, nt, steps, res['1'][0], res['1'][0] / steps * 1e6, res['0'][0], res['0'][0] / steps * 1e6,
         bool(np.isfinite(a).all()), np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)))

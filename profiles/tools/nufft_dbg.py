import sys, os, io, contextlib
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import synth
from impdar_amd.lib.RadarData import RadarData
snum, tnum = 2100, 16
geo = synth.geometry(snum, tnum)
data = (synth.noise_radargram(snum, tnum, seed=snum) + 0.5).astype(np.float32)
out = {}
for mode in ('6', '0'):
    os.environ['IMPDAR_PS_MFMA'] = mode
    d = RadarData(None); d.data, d.snum, d.tnum = data.copy(), snum, tnum
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    with contextlib.redirect_stdout(io.StringIO()):
        d.migrate('phsh', vel=1.69e8, htaper=20, vtaper=30)
    out[mode] = d.data
e = np.abs(out['6'] - out['0']).max(axis=1) / np.abs(out['0']).max()
for lo in range(0, snum, 150):
    print(lo, float(e[lo:lo+150].max()))

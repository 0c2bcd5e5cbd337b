"""Three constant-velocity and three v(z) calls of the config-5 phase shift through the library given as argv[1]
(timing-only ablation builds of ps_mfma_kernel: profiles/tools/variant_build.sh, r03_runs/r03_g37.sh)."""
import os, sys, io, contextlib
os.environ['IMPDAR_HIP_LIB'] = sys.argv[1]
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData
n = 8192
geo = synth.geometry(n, n)
x = np.random.default_rng(0).standard_normal((n, n)).astype(np.float32)
Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
tab = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
for vel in (1.69e8, tab):
    for i in range(3):
        d = RadarData(None)
        d.data, (d.snum, d.tnum) = x, x.shape
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        d.to_device()
        with contextlib.redirect_stdout(io.StringIO()):
            d.migrate('phsh', vel=vel, htaper=100, vtaper=1000)
        d._dev.free(); d._dev = None

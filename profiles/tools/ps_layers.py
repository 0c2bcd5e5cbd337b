"""Device ms of the float32 phase shift against the number of layers of the (v, z) table (layers of equal thickness,
velocity rising from 1.69e8 to 2.2e8), resident.  usage: ps_layers.py [n]"""
import sys, os, json, io, contextlib
import ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData
lib, ctx = _hip.load(), _hip.context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
geo = synth.geometry(n, n)
x = np.random.default_rng(0).standard_normal((n, n)).astype(np.float32)
Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
for nl in (1, 2, 3, 4, 6, 8, 12, 20, 40, 80):
    z = np.linspace(0., 2.0 * Rp, nl + 1)
    v = np.linspace(1.69e8, 2.2e8, nl + 1)
    tab = np.stack([v, z], axis=1) if nl > 1 else np.array([[1.69e8, 0.], [1.69e8, 2.0 * Rp]])
    ms = []
    for i in range(3):
        d = RadarData(None)
        d.data, (d.snum, d.tnum) = x, x.shape
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        d.to_device()
        with contextlib.redirect_stdout(io.StringIO()):
            d.migrate('phsh', vel=tab, htaper=100, vtaper=1000)
        vv = C.c_float(); _hip.check(lib.impdar_ctx_last_ms(ctx, C.byref(vv))); ms.append(round(vv.value, 2))
        d._dev.free(); d._dev = None
    print(json.dumps({'n': n, 'table rows': nl + 1, 'device_ms': ms}), flush=True)

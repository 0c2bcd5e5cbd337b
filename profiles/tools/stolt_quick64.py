"""Device ms of the Stolt migration on float64 and float32 data at 4096^2, resident."""
import sys, os, json, io, contextlib
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData
lib, ctx = _hip.load(), _hip.context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
geo = synth.geometry(n, n)
rng = np.random.default_rng(0)
out = {}
for dt in (np.float32, np.float64):
    x = rng.standard_normal((n, n)).astype(dt)
    ms = []
    for i in range(4):
        d = RadarData(None)
        d.data, (d.snum, d.tnum) = x, x.shape
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        d.to_device()
        with contextlib.redirect_stdout(io.StringIO()):
            d.migrate('stolt', htaper=100, vtaper=1000)
        v = C.c_float()
        _hip.check(lib.impdar_ctx_last_ms(ctx, C.byref(v)), 'last_ms')
        ms.append(v.value)
        d._dev.free(); d._dev = None
    out[np.dtype(dt).name] = round(float(np.median(ms[1:])), 4)
print(json.dumps(out))

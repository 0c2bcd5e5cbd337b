"""Config-2 Stolt migration (4096 x 4096 float32, resident), a few calls (for a rocprofv3 timeline)."""
import sys, os, io, contextlib
import ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData
lib, ctx = _hip.load(), _hip.context()
n = 4096
geo = synth.geometry(n, n)
x = np.random.default_rng(0).standard_normal((n, n)).astype(np.float32)
for i in range(4):
    d = RadarData(None)
    d.data, (d.snum, d.tnum) = x, x.shape
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    d.to_device()
    with contextlib.redirect_stdout(io.StringIO()):
        d.migrate('stolt', htaper=100, vtaper=1000)
    v = C.c_float()
    _hip.check(lib.impdar_ctx_last_ms(ctx, C.byref(v)), 'impdar_ctx_last_ms')
    print('device_ms %.3f' % v.value, flush=True)
    d._dev.free(); d._dev = None

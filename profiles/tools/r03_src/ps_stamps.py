"""Diagnostics (build/diag/lib_stamps.so, -DIMPDAR_PS_DIAG_STAMPS): core-clock stamps inside one workgroup of the config-5
rotate-accumulate kernels -- where a 16-step tile's time goes (body, reduce-scatter, barrier, tail), waves 0 and 7."""
import sys, os, io, contextlib
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData

lib, ctx = _hip.load(), _hip.context()
n = 8192
rng = np.random.default_rng(0)
geo = synth.geometry(n, n)
x = rng.standard_normal((n, n)).astype(np.float32)
Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
flat = np.array([[1.69e8, 0.], [1.69e8, 10. * Rp]])
raw = C.CDLL(_hip.LIB_PATH)
for name, vel in (('const', 1.69e8), ('vz_flat', flat)):
    d = RadarData(None)
    d.data, (d.snum, d.tnum) = x, x.shape
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    d.to_device()
    with contextlib.redirect_stdout(io.StringIO()):
        d.migrate('phsh', vel=vel, htaper=100, vtaper=1000)
    buf = (C.c_longlong * 160)()
    assert raw.impdar_ps_debug_stamps(buf) == 0
    st = np.array(buf[:], dtype=np.int64).reshape(2, 16, 5)
    print(name)
    for w in (0, 1):
        dt = np.diff(st[w], axis=1)                       # body, reduce, barrier, tail
        nxt = st[w, 1:, 0] - st[w, :-1, 4]                # loop back edge
        per = st[w, 1:, 0] - st[w, :-1, 0]
        print('  wave %d: tile period %s' % (0 if w == 0 else 7, per[:8]))
        print('          body %s reduce %s barrier %s tail %s backedge %s' % (dt[:6, 0], dt[:6, 1], dt[:6, 2], dt[:6, 3], nxt[:6]))
    d._dev.free()

"""Device / kernel ms of the float32 phase shift at n x n for (v, z) tables of the given row counts (equal layers,
1.69e8 -> 2.2e8 m/s), resident.   usage: ps_table_quick.py n rows[,rows...] [reps]   (IMPDAR_PS_MFMA selects the path)"""
import sys, os, json, io, contextlib
import ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData
lib, ctx = _hip.load(), _hip.context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rows = [int(r) for r in (sys.argv[2] if len(sys.argv) > 2 else '41,81').split(',')]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
geo = synth.geometry(n, n)
x = np.random.default_rng(0).standard_normal((n, n)).astype(np.float32)
Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
for nr in rows:
    tab = np.stack([np.linspace(1.69e8, 2.2e8, nr), np.linspace(0., 2.0 * Rp, nr)], axis=1)
    ms, kms, kern = [], [], ''
    for i in range(reps):
        d = RadarData(None)
        d.data, (d.snum, d.tnum) = x, x.shape
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        d.to_device()
        with contextlib.redirect_stdout(io.StringIO()):
            d.migrate('phsh', vel=tab, htaper=100, vtaper=1000)
        v = C.c_float(); _hip.check(lib.impdar_ctx_last_ms(ctx, C.byref(v))); ms.append(round(v.value, 2))
        _hip.check(lib.impdar_ctx_last_kernel_ms(ctx, C.byref(v))); kms.append(round(v.value, 2))
        buf = C.create_string_buffer(1024); _hip.check(lib.impdar_ctx_last_metrics(ctx, buf, len(buf))); mt = json.loads(buf.value.decode()); kern = mt['kernel']; nl = mt.get('long_runs')
        d._dev.free(); d._dev = None
    print(json.dumps({'n': n, 'table rows': nr, 'long_runs': nl, 'kernel': kern, 'device_ms': ms, 'kernel_ms': kms}), flush=True)

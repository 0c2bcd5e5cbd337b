"""Device ms of the phase shift against the number of layers of the (v, z) table, on the transform path (IMPDAR_PS_MFMA=6: every
table through ps_nufft_kernel) and on what takes tables of many layers otherwise (=3 float32: ps_runs_kernel; =0 float64: ps_vz64_kernel):
where the library should change over.  usage: r06_layers.py [n] [float32|float64]   (one process per setting: the knob is read per call)"""
import sys, os, json, io, contextlib
import ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData
lib, ctx = _hip.load(), _hip.context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dtype = np.dtype(sys.argv[2]) if len(sys.argv) > 2 else np.dtype('float32')
geo = synth.geometry(n, n)
x = np.random.default_rng(0).standard_normal((n, n)).astype(dtype)
Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
for nl in (3, 7, 11, 15, 19, 23, 31, 40):
    z = np.linspace(0., 2.0 * Rp, nl + 1)
    v = np.linspace(1.69e8, 2.2e8, nl + 1)
    tab = np.stack([v, z], axis=1)
    row = {'n': n, 'dtype': str(dtype), 'table rows': nl + 1}
    for mode in ('6', '3' if dtype == np.float32 else '0', '1'):
        os.environ['IMPDAR_PS_MFMA'] = mode
        ms = []
        for i in range(3):
            d = RadarData(None)
            d.data, (d.snum, d.tnum) = x, x.shape
            d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
            d.to_device()
            with contextlib.redirect_stdout(io.StringIO()):
                d.migrate('phsh', vel=tab, htaper=100, vtaper=1000)
            vv = C.c_float(); _hip.check(lib.impdar_ctx_last_ms(ctx, C.byref(vv))); ms.append(round(vv.value, 2))
            buf = C.create_string_buffer(1024); _hip.check(lib.impdar_ctx_last_metrics(ctx, buf, len(buf))); kern = json.loads(buf.value.decode())['kernel']
            d._dev.free(); d._dev = None
        row['mode %s' % mode] = [kern, sorted(ms)[1]]
    print(json.dumps(row), flush=True)

"""Device ms of the config-5 phase-shift migration (8192x8192 float32, resident), constant velocity and 1-D v(z)
table (same-box A/B of the rotate-accumulate kernels; run under rocprofv3 --kernel-trace --stats for kernel times)."""
import sys, os, json, io, time, contextlib
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData

lib, ctx = _hip.load(), _hip.context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(0)
geo = synth.geometry(n, n)
x = rng.standard_normal((n, n)).astype(np.float32)
Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
tab = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
out = {}
flat = np.array([[1.69e8, 0.], [1.69e8, 10. * Rp]])      # one constant-velocity run through the v(z) kernel
for name, vel in (('const', 1.69e8), ('vz', tab), ('vz_flat', flat)):
    ms, kms = [], []
    for i in range(reps + 1):
        d = RadarData(None)
        d.data, (d.snum, d.tnum) = x, x.shape
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        d.to_device()
        with contextlib.redirect_stdout(io.StringIO()):
            d.migrate('phsh', vel=vel, htaper=100, vtaper=1000)
        v = C.c_float()
        _hip.check(lib.impdar_ctx_last_ms(ctx, C.byref(v)), 'impdar_ctx_last_ms')
        ms.append(v.value)
        _hip.check(lib.impdar_ctx_last_kernel_ms(ctx, C.byref(v)), 'impdar_ctx_last_kernel_ms')
        kms.append(v.value)
        buf = C.create_string_buffer(1024); _hip.check(lib.impdar_ctx_last_metrics(ctx, buf, len(buf))); kern = json.loads(buf.value.decode())['kernel']
        d._dev.free()
        d._dev = None
    out[name] = {'kernel': kern, 'device_ms': round(float(np.median(ms[1:])), 2), 'kernel_ms': round(float(np.median(kms[1:])), 2), 'all': [round(m, 2) for m in ms]}
print(json.dumps(out))

"""Device ms of the phase shift with a velocity that changes at EVERY depth step (a linear gradient: no runs of constant
velocity -- ps_smooth_kernel since round 4, the per-step kernels before), float32 and float64, resident.
usage: ps_smooth.py [n]"""
import sys, os, json
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from impdar_amd import _hip, synth
lib, ctx = _hip.load(), _hip.context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
geo = synth.geometry(n, n)
nt = n
kx = 2. * np.pi * np.fft.fftfreq(n, d=1.0)
ws = 2. * np.pi * np.fft.fftfreq(nt, d=geo['dt'])
vm = np.linspace(1.69e8, 2.2e8, n)
p = lambda a: _hip.as_dp(a)[1]
out = {}
for dt in (np.float32, np.float64):
    x = np.random.default_rng(0).standard_normal((n, n)).astype(dt)
    d_in = _hip.DeviceArray.from_host(ctx, x)
    d_out = _hip.DeviceArray(ctx, (n, n), dt)
    ms, kms = [], []
    for i in range(3):
        _hip.check(lib.impdar_phaseshift_dev(ctx, d_in.ptr, _hip.dtype_code(dt), n, n, nt, p(kx), p(ws), geo['dt'], p(geo['travel_time']),
                                             0.0, p(vm), n, 100.0, 1000.0, d_out.ptr), 'ps')
        v = C.c_float(); _hip.check(lib.impdar_ctx_last_ms(ctx, C.byref(v))); ms.append(round(v.value, 2))
        _hip.check(lib.impdar_ctx_last_kernel_ms(ctx, C.byref(v))); kms.append(round(v.value, 2))
    buf = C.create_string_buffer(1024)
    lib.impdar_ctx_last_metrics(ctx, buf, len(buf))
    out[np.dtype(dt).name] = dict(device_ms=ms, kernel_ms=kms, kernel=json.loads(buf.value.decode()).get('kernel'),
                                  finite=bool(np.isfinite(d_out.to_host()).all()))
    d_in.free(); d_out.free()
print(json.dumps({'n': n, 'linear gradient v(z)': out}))

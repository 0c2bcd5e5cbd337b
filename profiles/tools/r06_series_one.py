import sys, os
sys.path.insert(0, os.getcwd())
os.environ['IMPDAR_PS_MFMA'] = '7'
import numpy as np, ctypes as C
from impdar_amd import _hip, synth
lib, ctx = _hip.load(), _hip.context()
n = 8192
geo = synth.geometry(n, n)
kx = 2. * np.pi * np.fft.fftfreq(n, d=1.0); ws = 2. * np.pi * np.fft.fftfreq(n, d=geo['dt'])
p = lambda a: _hip.as_dp(a)[1]
vm = np.ascontiguousarray(1.69e8 + 0.5e8 * np.linspace(0., 1., n))
x = np.random.default_rng(0).standard_normal((n, n)).astype(np.float32)
d_in = _hip.DeviceArray.from_host(ctx, x); d_out = _hip.DeviceArray(ctx, (n, n), np.float32)
for i in range(2):
    _hip.check(lib.impdar_phaseshift_dev(ctx, d_in.ptr, 0, n, n, n, p(kx), p(ws), geo['dt'], p(geo['travel_time']), 0.0, p(vm), n, 100.0, 1000.0, d_out.ptr), 'ps')

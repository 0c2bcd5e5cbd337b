"""Wall time of DeviceArray.to_host_f64 / to_host for float32 and float64 device arrays of config-3 size (the result
download of the resident chain)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import _hip
lib, ctx = _hip.load(), _hip.context()
for dt in (np.float32, np.float64):
    a = np.random.default_rng(0).standard_normal((4096, 10000)).astype(dt)
    d = _hip.DeviceArray.from_host(ctx, a)
    for what in ('to_host_f64', 'to_host'):
        for i in range(3):
            t0 = time.perf_counter()
            out = getattr(d, what)()
            t = (time.perf_counter() - t0) * 1e3
            print('%s %s: %.2f ms (%.1f GB/s of device bytes)' % (np.dtype(dt).name, what, t, a.nbytes / t / 1e6), flush=True)
            del out
    d.free()

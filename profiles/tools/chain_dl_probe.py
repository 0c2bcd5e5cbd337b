"""Why does the result download of the resident chain run at 29 GB/s?  The chain's own from_device against downloads of
the same device array again, after a pause, and of a fresh device array of the same shape."""
import sys, time, io, contextlib
import numpy as np
sys.path.insert(0, '.')
from impdar_amd import _hip
from impdar_amd.lib.NoInitRadarData import NoInitRadarDataFiltering
lib = _hip.load(); ctx = _hip.context()
snum, tnum = 4096, 10000
rng = np.random.default_rng(0)
x = rng.standard_normal((snum, tnum)).astype(np.float32)
dist = np.hstack(([0.], np.cumsum(0.6 + 0.8 * rng.random(tnum - 1)))) / 1000.
def T(f):
    lib.impdar_ctx_sync(ctx); t0 = time.perf_counter(); r = f(); lib.impdar_ctx_sync(ctx); return (time.perf_counter() - t0) * 1e3, r
for rep in range(3):
    d = NoInitRadarDataFiltering(); d.data, (d.snum, d.tnum) = x.copy(), x.shape; d.dt, d.dist = 1e-8, dist.copy()
    d.travel_time = np.arange(snum) * 1e-2
    for a in ['lat', 'long', 'x_coord', 'y_coord', 'decday', 'pressure', 'elev']: setattr(d, a, np.arange(tnum, dtype=float))
    d.trig = np.zeros(tnum)
    with contextlib.redirect_stdout(io.StringIO()):
        d.to_device(); d.vertical_band_pass(2., 10.); d.constant_space(1.0); d.migrate('stolt', htaper=100, vtaper=1000)
    dev = d._dev
    t1, o1 = T(dev.to_host)
    t2, o2 = T(dev.to_host)
    time.sleep(0.2)
    t3, o3 = T(dev.to_host)
    fresh = _hip.DeviceArray.from_host(ctx, o1)
    t4, o4 = T(fresh.to_host)
    pre = np.empty_like(o1); pre[:] = 0
    t5, _ = T(lambda: lib.impdar_dev_download(ctx, pre.ctypes.data, dev.ptr, dev.nbytes))
    print('shape %s %s: chain array %.1f ms, again %.1f, after a pause %.1f, fresh device array %.1f, into touched host memory %.1f' % (dev.shape, dev.dtype, t1, t2, t3, t4, t5), flush=True)
    fresh.free(); d.from_device()

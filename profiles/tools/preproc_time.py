#!/usr/bin/env python3
"""Device time of the resident band pass and re-spacing kernels at BASELINE config-3 size (4096 x 10000),
HIP-synchronised wall clock over 10 launches each."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from impdar_amd import _hip, preproc
_hip.load(); ctx = _hip.context(0); lib = _hip.load()
snum, tnum = 4096, 10000
rng = np.random.default_rng(0)
for dtype in (np.float32, np.float64):
    data = rng.standard_normal((snum, tnum)).astype(dtype)
    d = _hip.DeviceArray.from_host(ctx, data)
    es = data.dtype.itemsize
    for label, spec in (('butter5 filtfilt', preproc.design_filter(1e-8, 2., 10.)),
                        ('fir100', preproc.design_filter(1e-8, 2., 10., order=100, filttype='fir'))):
        preproc.filter_dev(d, spec); lib.impdar_ctx_sync(ctx)
        t0 = time.perf_counter()
        for _ in range(10): preproc.filter_dev(d, spec)
        lib.impdar_ctx_sync(ctx); ms = (time.perf_counter() - t0) / 10 * 1e3
        if spec[0] == 'iir':
            L = snum + 6 * len(spec[1])
            algo = snum * tnum * es * 2 + L * tnum * 8 * 2      # read x, write y_fwd, read y_fwd, write out
        else:
            algo = snum * tnum * es * 4                        # read, write scratch, copy back (read + write)
        print('%s %-18s %.3f ms  %.0f GB/s algorithmic' % (np.dtype(dtype).name, label, ms, algo / ms / 1e6))
    d.free()
    d = _hip.DeviceArray.from_host(ctx, data)
    dist = np.hstack(([0.], np.cumsum(0.6 + 0.8 * rng.random(tnum - 1)))) / 1000.
    plan = preproc.SpacingPlan(dist, 1.0)
    o = plan.apply_dev(d); lib.impdar_ctx_sync(ctx); o.free()
    t0 = time.perf_counter()
    for _ in range(10):
        o = plan.apply_dev(d); lib.impdar_ctx_sync(ctx); o.free()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    algo = snum * tnum * es + snum * plan.n_new * 8
    print('%s %-18s %.3f ms  %.0f GB/s algorithmic (incl. alloc + table upload), %d -> %d traces'
          % (np.dtype(dtype).name, 'constant_space', ms, algo / ms / 1e6, tnum, plan.n_new))
    d.free()

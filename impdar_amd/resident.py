"""Keeping a radargram in HBM across the steps of a processing chain.

``dat.to_device()`` uploads ``dat.data`` once; ``vertical_band_pass``, ``constant_space`` and
``migrate('kirch' | 'stolt' | 'phsh')`` then work on the resident array through the ``*_dev`` entry points of
the C ABI (no PCIe traffic between steps), and ``dat.from_device()`` brings the result back.  While resident,
``dat.data`` is None.  What has no resident form ('tk', the 2-D v(x, z) branch of 'phsh', SeisUnix)
round-trips through the host.
"""
import os

import numpy as np

from . import _hip
from .kirchhoff import KirchhoffPlan


def to_device(self):
    """Upload ``self.data`` (float32 / float64) to the GPU and keep it there."""
    if getattr(self, '_dev', None) is not None:
        return self
    data = np.asarray(self.data)
    if data.dtype not in (np.float32, np.float64):
        raise TypeError('only float32 / float64 radargrams can be held on the device, got %s' % data.dtype)
    if data.ndim != 2:
        raise ValueError('The input array must be of size (snum, tnum)')
    self._dev = _hip.DeviceArray.from_host(_hip.context(), data)
    self._dev_widen = False
    self.data = None
    return self


def from_device(self):
    """Download the resident radargram into ``self.data`` and release the device copy."""
    dev = getattr(self, '_dev', None)
    if dev is None:
        return self
    if getattr(self, '_dev_widen', False):
        out = dev.to_host_f64()            # migrationKirchhoff / PhaseShift always hand back float64 (mig_python.py:118, :282)
    else:
        out = dev.to_host()
    dev.free()
    self._dev = None
    self._dev_widen = False
    self.data = out
    return self


def kirchhoff_resident(dat, vel=1.69e8, nearfield=False):
    """migrationKirchhoff (mig_python.py:63-123) on the resident array; same kernel choice as the host-buffer
    form (float32 on uniform grids -> LDS-ring kernel, else the fp64 reference-order kernels)."""
    dev = dat._dev
    if dev.shape != (dat.snum, dat.tnum):
        raise ValueError('The input array must be of size (snum, tnum)')
    print('Kirchhoff Migration (diffraction summation) of %.0fx%.0f matrix' % (dat.snum, dat.tnum))
    print('Using the MI355X HIP engine (resident)')
    mode = os.environ.get('IMPDAR_KIRCH_MODE') or 'auto'
    if mode not in ('auto', 'exact', 'fast'):
        raise ValueError('mode must be one of auto, exact, fast')
    if mode == 'fast' and dev.dtype != np.float32:
        # explicit opt-in to the float32 kernel for float64 data (as migrationKirchhoff(mode='fast') on the host)
        d32 = _hip.DeviceArray(dev.ctx, dev.shape, np.float32)
        _hip.check(_hip.load().impdar_cast_dev(dev.ctx, dev.ptr, _hip.dtype_code(dev.dtype), d32.ptr, _hip.F32,
                                               int(np.prod(dev.shape))), 'impdar_cast_dev')
        _hip.check(_hip.load().impdar_ctx_sync(dev.ctx), 'impdar_ctx_sync')
        dev.free()
        dev = dat._dev = d32
    plan = KirchhoffPlan(dev.ctx, dev.dtype, dat.snum, dat.tnum, dat.dist, dat.travel_time, vel, nearfield, mode)
    d_out = _hip.DeviceArray(dev.ctx, dev.shape, dev.dtype)
    try:
        plan.prep(dev, dat.tnum, 0, dat.tnum)
        plan.migrate(d_out, 0, dat.tnum)
        plan.sync()
    except Exception:
        d_out.free()
        raise
    finally:
        plan.destroy()
    dev.free()
    dat._dev = d_out
    dat._dev_widen = True
    return dat


def stolt_resident(dat, vel=1.68e8, htaper=100, vtaper=1000):
    """migrationStolt (mig_python.py:126-208) on the resident array."""
    from .lib.migrationlib.mig_hip import _kx
    dev = dat._dev
    if dev.shape != (dat.snum, dat.tnum):
        raise ValueError('The input array must be of size (snum, tnum)')
    print('Stolt Migration (f-k migration) of %.0fx%.0f matrix' % (dat.snum, dat.tnum))
    ws = 2. * np.pi * np.fft.rfftfreq(dat.snum, d=dat.dt)
    kx = _kx(dat)
    nout = 2 * (dat.snum // 2)
    d_out = _hip.DeviceArray(dev.ctx, (nout, dat.tnum), dev.dtype)
    _, p_kx = _hip.as_dp(kx)
    _, p_ws = _hip.as_dp(ws)
    rc = _hip.load().impdar_stolt_dev(dev.ctx, dev.ptr, _hip.dtype_code(dev.dtype), dat.snum, dat.tnum, p_kx, p_ws,
                                      float(vel), float(htaper), float(vtaper), d_out.ptr)
    if rc:
        d_out.free()
    _hip.check(rc, 'impdar_stolt')
    _hip.check(_hip.load().impdar_ctx_sync(dev.ctx), 'impdar_ctx_sync')
    dev.free()
    dat._dev = d_out
    return dat

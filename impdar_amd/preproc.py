"""Host side of the processing steps that sit in front of a migration: ``vertical_band_pass`` and
``constant_space``.  Filter design and the O(tnum) geometry stay on the host (SciPy / NumPy,
as in the reference); everything that touches the (snum, tnum) radargram runs in ``csrc/preproc.hip`` through
the C ABI, on host buffers or on an array that is already resident in HBM.

Reference: ``src/impdar/lib/RadarData/_RadarDataFiltering.py:469-549`` and
``src/impdar/lib/RadarData/_RadarDataProcessing.py:499-583``.
"""
import ctypes as C

import numpy as np

from . import _hip


# ------------------------------------------------------------------------------------------------ band pass
def design_filter(dt, low, high, order=5, filttype='butter', cheb_rp=5):
    """('iir', b, a, zi) or ('fir', taps) for the reference's corner frequencies (:511-540).  The reference
    accepts ``fir_window`` but never hands it to ``firwin``; neither does this."""
    from scipy import signal
    nyquist_freq = 0.5 * (1.0 / dt)
    corner_freq = np.zeros((2,))
    corner_freq[0] = low * 1.0e6 / nyquist_freq
    corner_freq[1] = high * 1.0e6 / nyquist_freq
    ft = filttype.lower()
    if ft in ['butter', 'butterworth']:
        b, a = signal.butter(order, corner_freq, 'bandpass')
    elif ft in ['cheb', 'chebyshev']:
        b, a = signal.cheby1(order, cheb_rp, corner_freq, 'bandpass')
    elif ft == 'bessel':
        b, a = signal.bessel(order, corner_freq, 'bandpass')
    elif ft == 'fir':
        return ('fir', np.ascontiguousarray(signal.firwin(order + 1, corner_freq, pass_zero=False), dtype=np.float64))
    else:
        raise ValueError('Filter type {:s} is not recognized'.format(filttype))
    # filtfilt's steady-state initial conditions: SciPy solves an ill-conditioned linear system with LAPACK,
    # so the same call is the only way to the same numbers
    zi = signal.lfilter_zi(b, a)
    n = max(len(a), len(b))
    b = np.r_[b, np.zeros(n - len(b))]
    a = np.r_[a, np.zeros(n - len(a))]
    return ('iir', np.ascontiguousarray(b, dtype=np.float64), np.ascontiguousarray(a, dtype=np.float64),
            np.ascontiguousarray(zi, dtype=np.float64))


def _run_filter(lib, ctx, ptr, code, snum, tnum, spec, dev):
    if spec[0] == 'iir':
        _, b, a, zi = spec
        fn = lib.impdar_filtfilt_dev if dev else lib.impdar_filtfilt
        rc = fn(ctx, ptr, code, snum, tnum, _hip.as_dp(b)[1], _hip.as_dp(a)[1], len(b), _hip.as_dp(zi)[1])
        _hip.check(rc, 'impdar_filtfilt')
    else:
        taps = spec[1]
        fn = lib.impdar_fir_shift_dev if dev else lib.impdar_fir_shift
        _hip.check(fn(ctx, ptr, code, snum, tnum, _hip.as_dp(taps)[1], len(taps)), 'impdar_fir_shift')


def filter_host(data, spec):
    """Filtered copy of a host radargram in its own dtype (integers: computed in float64, then ``astype``
    as the reference's ``filtfilt(...).astype(self.data.dtype)``)."""
    data = np.asarray(data)
    if data.ndim != 2:
        raise ValueError('data must be (snum, tnum)')
    if np.iscomplexobj(data):
        raise TypeError('vertical_band_pass on complex data is not supported by the MI355X engine')
    work = np.array(data, dtype=data.dtype if data.dtype in (np.float32, np.float64) else np.float64, order='C')
    snum, tnum = work.shape
    _run_filter(_hip.load(), _hip.context(), work.ctypes.data_as(C.c_void_p), _hip.dtype_code(work.dtype), snum, tnum,
                spec, dev=False)
    return work.astype(data.dtype) if work.dtype != data.dtype else work


def filter_dev(d_arr, spec):
    """In-place on a resident :class:`impdar_amd._hip.DeviceArray` (float32 / float64)."""
    snum, tnum = d_arr.shape
    _run_filter(_hip.load(), d_arr.ctx, d_arr.ptr, _hip.dtype_code(d_arr.dtype), snum, tnum, spec, dev=True)


# ------------------------------------------------------------------------------------------ constant spacing
def interp1d_linear(x, y, x_new):
    """``scipy.interpolate.interp1d(x, y)(x_new)`` for the small per-trace attribute vectors: sorted
    abscissae, range check, ``np.interp`` for float64/int 1-D values and the slope form otherwise."""
    x = np.asarray(x)
    y = np.asarray(y)
    if not np.issubdtype(y.dtype, np.inexact):
        y = y.astype(np.float64)
    ind = np.argsort(x, kind='mergesort')
    x = x[ind]
    y = np.take(y, ind, axis=-1)
    _check_range(x, x_new)
    np_dtypes = (np.dtype(np.float64), np.dtype(np.int_))
    if y.ndim == 1 and x.dtype in np_dtypes and y.dtype in np_dtypes:
        return np.interp(x_new, x, y)
    idx = np.searchsorted(x, x_new).clip(1, len(x) - 1).astype(int)
    yt = np.moveaxis(y, -1, 0)
    shp = (-1,) + (1,) * (yt.ndim - 1)
    slope = (yt[idx] - yt[idx - 1]) / (x[idx] - x[idx - 1]).reshape(shp)
    return np.moveaxis(slope * (x_new - x[idx - 1]).reshape(shp) + yt[idx - 1], 0, -1)


def _check_range(x, x_new):
    x_new = np.asarray(x_new)
    if x_new.size == 0:
        return
    below = x_new < x[0]
    above = x_new > x[-1]
    if below.any():
        raise ValueError("A value ({}) in x_new is below the interpolation range's minimum value ({})."
                         .format(x_new[np.argmax(below)], x[0]))
    if above.any():
        raise ValueError("A value ({}) in x_new is above the interpolation range's maximum value ({})."
                         .format(x_new[np.argmax(above)], x[-1]))


class SpacingPlan(object):
    """The O(tnum) geometry of ``constant_space`` (:530-547) and the gather tables of its interpolation."""

    def __init__(self, dist_km, spacing, min_movement=1.0e-2):
        dist = dist_km            # corrected in place, as the reference corrects self.dist
        self.good_vals = np.hstack((np.array([True]), np.diff(dist * 1000.) >= min_movement))
        for i in np.flatnonzero(~self.good_vals):
            dist[i:] = dist[i:] - (dist[i] - dist[i - 1])
        self.temp_dist = dist[self.good_vals]
        self.new_dists = np.arange(np.min(self.temp_dist), np.max(self.temp_dist), step=spacing / 1000.0)
        self.n_new = len(self.new_dists)
        if self.n_new == 0:
            # nothing moved (a single usable shot): interp1d evaluated at no points, an empty radargram
            self.lo = self.hi = np.zeros(0, dtype=np.int32)
            self.den = self.t = np.zeros(0, dtype=np.float64)
            return
        order = np.argsort(self.temp_dist, kind='mergesort')
        xs = self.temp_dist[order]
        cols = np.flatnonzero(self.good_vals)[order]
        _check_range(xs, self.new_dists)
        idx = np.searchsorted(xs, self.new_dists).clip(1, len(xs) - 1).astype(int)
        self.lo = np.ascontiguousarray(cols[idx - 1], dtype=np.int32)
        self.hi = np.ascontiguousarray(cols[idx], dtype=np.int32)
        self.den = np.ascontiguousarray(xs[idx] - xs[idx - 1], dtype=np.float64)
        self.t = np.ascontiguousarray(self.new_dists - xs[idx - 1], dtype=np.float64)

    def _tables(self):
        ip = C.POINTER(C.c_int)
        return (self.lo.ctypes.data_as(ip), self.hi.ctypes.data_as(ip), _hip.as_dp(self.den)[1], _hip.as_dp(self.t)[1])

    def apply_host(self, data):
        """float64 (snum, n_new) interpolation of a host radargram (complex: real and imaginary parts)."""
        data = np.asarray(data)
        if np.iscomplexobj(data):
            return self.apply_host(np.ascontiguousarray(data.real)) + 1.j * self.apply_host(np.ascontiguousarray(data.imag))
        if data.dtype not in (np.float32, np.float64):
            data = data.astype(np.float64)
        data = np.ascontiguousarray(data)
        snum, tnum = data.shape
        out = np.empty((snum, self.n_new), dtype=np.float64)
        if self.n_new == 0:
            return out
        lo, hi, den, t = self._tables()
        rc = _hip.load().impdar_trace_lerp(_hip.context(), data.ctypes.data_as(C.c_void_p), _hip.dtype_code(data.dtype),
                                          snum, tnum, lo, hi, den, t, self.n_new, out.ctypes.data_as(C.c_void_p))
        _hip.check(rc, 'impdar_trace_lerp')
        return out

    def apply_dev(self, d_arr):
        """New resident float64 (snum, n_new) array; the caller frees the old one."""
        snum, tnum = d_arr.shape
        d_out = _hip.DeviceArray(d_arr.ctx, (snum, max(self.n_new, 1)), np.float64)
        d_out.shape = (snum, self.n_new)
        d_out.nbytes = snum * self.n_new * 8
        if self.n_new:
            lo, hi, den, t = self._tables()
            rc = _hip.load().impdar_trace_lerp_dev(d_arr.ctx, d_arr.ptr, _hip.dtype_code(d_arr.dtype), snum, tnum,
                                                  lo, hi, den, t, self.n_new, d_out.ptr)
            _hip.check(rc, 'impdar_trace_lerp')
        return d_out

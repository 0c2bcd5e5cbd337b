"""CPU oracle for the two processing steps in front of a migration.

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path.

Restates, in NumPy, what the reference computes in

* ``RadarData.vertical_band_pass`` (``src/impdar/lib/RadarData/_RadarDataFiltering.py:469-549``), which is
  ``scipy.signal.filtfilt`` / ``scipy.signal.lfilter`` on coefficients from ``scipy.signal.butter`` /
  ``cheby1`` / ``bessel`` / ``firwin``;
* ``RadarData.constant_space`` (``src/impdar/lib/RadarData/_RadarDataProcessing.py:499-583``), which is
  ``scipy.interpolate.interp1d`` (linear).

SciPy is a third-party dependency of the reference (``scipy>0.19.0``, unpinned; 1.15.3 in the build container).
The filter *design* functions are called from SciPy here as the reference does; what is restated is the
arithmetic applied to the radargram: the published algorithms of ``filtfilt`` (odd extension by
``3*max(len(a), len(b))`` samples, ``lfilter_zi`` steady-state initial conditions, forward pass, reversed
pass), of ``lfilter`` (transposed direct form II, coefficients normalised by ``a[0]``) and of ``interp1d``'s
linear branch (slope form on the bracketing knots found by ``searchsorted``).

Parity pinned by ``tests/golden/V*_vbp_*.npz`` and ``C*_cspace_*.npz`` (outputs of the reference itself,
``tests/golden/make_golden.py``).
"""
import numpy as np


def design(dt, low, high, order=5, filttype='butter', cheb_rp=5):
    """(kind, b, a) with kind 'iir' or 'fir'; corner frequencies as the reference forms them (:511-522)."""
    from scipy import signal
    nyquist = 0.5 * (1.0 / dt)
    corner = np.zeros((2,))
    corner[0] = low * 1.0e6 / nyquist
    corner[1] = high * 1.0e6 / nyquist
    ft = filttype.lower()
    if ft in ('butter', 'butterworth'):
        b, a = signal.butter(order, corner, 'bandpass')
        return 'iir', b, a
    if ft in ('cheb', 'chebyshev'):
        b, a = signal.cheby1(order, cheb_rp, corner, 'bandpass')
        return 'iir', b, a
    if ft == 'bessel':
        b, a = signal.bessel(order, corner, 'bandpass')
        return 'iir', b, a
    if ft == 'fir':
        return 'fir', signal.firwin(order + 1, corner, pass_zero=False), np.ones(1)
    raise ValueError('Filter type {:s} is not recognized'.format(filttype))


def lfilter_zi(b, a):
    """Steady-state delays of the transposed direct form II for a unit step (scipy.signal.lfilter_zi)."""
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    if a[0] != 1.0:
        b = b / a[0]
        a = a / a[0]
    n = max(len(a), len(b))
    a = np.r_[a, np.zeros(n - len(a))]
    b = np.r_[b, np.zeros(n - len(b))]
    # SciPy solves (I - A^T) zi = B with LAPACK (np.linalg.solve); the system is badly conditioned for
    # high-order band-pass designs (the explicit back-substitution formula differs from it by 1e-7 relative
    # at order 5 and by percents at order 8), so the same solver on the same matrix is part of the contract
    companion_t = np.zeros((n - 1, n - 1))
    companion_t[:, 0] = -a[1:]
    companion_t[np.arange(n - 2), np.arange(1, n - 1)] = 1.0
    i_minus_a = np.eye(n - 1) - companion_t
    zi = np.linalg.solve(i_minus_a, b[1:] - a[1:] * b[0])
    return zi


def lfilter_tdf2(b, a, x, z):
    """Transposed direct form II along axis 0 of x (float64), all traces at once; returns y.  ``z`` is
    (ncoef-1, tnum) and is updated in place.  Operation order of SciPy's C loop."""
    b = np.asarray(b, dtype=np.float64) / a[0]
    a = np.asarray(a, dtype=np.float64) / a[0]
    nc = len(b)
    y = np.empty_like(x)
    for i in range(x.shape[0]):
        xn = x[i]
        yn = z[0] + b[0] * xn
        for n in range(nc - 2):
            z[n] = z[n + 1] + xn * b[n + 1] - yn * a[n + 1]
        z[nc - 2] = xn * b[nc - 1] - yn * a[nc - 1]
        y[i] = yn
    return y


def filtfilt(b, a, x):
    """scipy.signal.filtfilt(b, a, x, axis=0) (padtype 'odd', padlen 3*ntaps, method 'pad'); float64 out."""
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    ntaps = max(len(a), len(b))
    a = np.r_[a, np.zeros(ntaps - len(a))]
    b = np.r_[b, np.zeros(ntaps - len(b))]
    edge = 3 * ntaps
    x = np.asarray(x)
    if x.shape[0] <= edge:
        raise ValueError('The length of the input vector x must be greater than padlen, which is %d.' % edge)
    # odd extension in the data's own dtype, as NumPy evaluates 2*x[0] - x[edge:0:-1]
    left = 2 * x[0:1] - x[edge:0:-1]
    right = 2 * x[-1:] - x[-2:-(edge + 2):-1]
    ext = np.concatenate((left, x, right), axis=0)
    zi = lfilter_zi(b, a)
    x0 = ext[0].astype(np.float64)
    y = lfilter_tdf2(b, a, ext.astype(np.float64), zi[:, None] * x0[None, :])
    y0 = y[-1]
    y = lfilter_tdf2(b, a, y[::-1].copy(), zi[:, None] * y0[None, :])
    return y[::-1][edge:-edge]


def fir_shift(taps, x):
    """lfilter(taps, 1.0, x, axis=0)[order:] for the rows the reference assigns (:538-540)."""
    taps = np.asarray(taps, dtype=np.float64)
    order = len(taps) - 1
    x64 = np.asarray(x, dtype=np.float64)
    n = x.shape[0] - order
    if n <= 0:
        return np.zeros((0,) + x.shape[1:])
    out = np.zeros((n,) + x.shape[1:])
    for i in range(order + 1):
        out += taps[i] * x64[order - i:order - i + n]
    return out


def vertical_band_pass(data, dt, low, high, order=5, filttype='butter', cheb_rp=5):
    """New data array (same dtype as ``data``), :527-540."""
    kind, b, a = design(dt, low, high, order, filttype, cheb_rp)
    if kind == 'iir':
        return filtfilt(b, a, data).astype(data.dtype)
    out = data.copy()
    if data.shape[0] > order:
        out[:-order, :] = fir_shift(b, data).astype(data.dtype)
    return out


def constant_space_plan(dist_km, spacing, min_movement=1.0e-2):
    """Host-side geometry of constant_space (:530-547): returns (good_vals, dist corrected for the stationary
    shots, new_dists)."""
    dist = np.array(dist_km, copy=True)
    good_vals = np.hstack((np.array([True]), np.diff(dist * 1000.) >= min_movement))
    for i in np.flatnonzero(~good_vals):
        dist[i:] = dist[i:] - (dist[i] - dist[i - 1])
    temp_dist = dist[good_vals]
    new_dists = np.arange(np.min(temp_dist), np.max(temp_dist), step=spacing / 1000.0)
    return good_vals, dist, new_dists


def interp_linear(x, y, x_new):
    """scipy interp1d(x, y)(x_new), linear, along the last axis of y; 1-D float y goes through np.interp as
    SciPy does, everything else through the slope form."""
    x = np.asarray(x)
    y = np.asarray(y)
    if not np.issubdtype(y.dtype, np.inexact):
        y = y.astype(np.float64)
    ind = np.argsort(x, kind='mergesort')
    x = x[ind]
    y = np.take(y, ind, axis=-1)
    x_new = np.asarray(x_new)
    if x_new.size and (np.any(x_new < x[0]) or np.any(x_new > x[-1])):
        raise ValueError('A value in x_new is outside the interpolation range.')
    if y.ndim == 1 and x.dtype in (np.dtype(np.float64), np.dtype(np.int_)) and \
            y.dtype in (np.dtype(np.float64), np.dtype(np.int_)):
        return np.interp(x_new, x, y)
    yt = np.moveaxis(y, -1, 0)                 # interpolation axis first, as interp1d's _y
    idx = np.searchsorted(x, x_new).clip(1, len(x) - 1).astype(int)
    lo = idx - 1
    hi = idx
    shp = (-1,) + (1,) * (yt.ndim - 1)
    slope = (yt[hi] - yt[lo]) / (x[hi] - x[lo]).reshape(shp)
    y_new = slope * (x_new - x[lo]).reshape(shp) + yt[lo]
    return np.moveaxis(y_new, 0, -1)


def constant_space(data, dist_km, spacing, min_movement=1.0e-2):
    """(new data, new_dists, good_vals, corrected dist): the radargram part of constant_space (:549-553)."""
    good_vals, dist, new_dists = constant_space_plan(dist_km, spacing, min_movement)
    temp_dist = dist[good_vals]
    # the reference tests np.iscomplexobj(self.data.dtype), which is always False, and so hands complex data
    # to interp1d as it is; SciPy's linear branch on complex values equals the two real interpolations
    new = interp_linear(temp_dist, data[:, good_vals], new_dists)
    return new, new_dists, good_vals, dist

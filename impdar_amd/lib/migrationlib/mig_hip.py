"""Host side of the MI355X migration engine.

Each function keeps the signature, defaults, printed messages, in-place
mutation of ``dat`` and error types of its reference counterpart in
``src/impdar/lib/migrationlib/mig_python.py`` (cited per function) and hands
the arithmetic to hand-written HIP kernels through the C ABI in
``include/impdar_hip.h``.  The only NumPy work left on the host is O(snum) or
O(tnum) set-up (frequency axes, gradient coefficients, velocity profile).
"""
from __future__ import print_function

import ctypes as C
import os
import time

import numpy as np

from ... import _hip

_MODES = {None: _hip.KIRCH_AUTO, 'auto': _hip.KIRCH_AUTO, 'exact': _hip.KIRCH_EXACT, 'fast': _hip.KIRCH_FAST}


def _check_data_shape(dat):
    """mig_python.py:646-648."""
    if np.size(dat.data, 1) != dat.tnum or np.size(dat.data, 0) != dat.snum:
        raise ValueError('The input array must be of size (snum, tnum)')


def _device_data(data):
    """float32 stays float32; everything else is processed as float64, the
    dtype the reference's arithmetic promotes to."""
    data = np.asarray(data)
    if data.dtype == np.float32:
        return np.ascontiguousarray(data), _hip.F32
    return np.ascontiguousarray(data, dtype=np.float64), _hip.F64


def gradient_coefficients(x):
    """Coefficients of ``numpy.gradient(f, x, axis=0)`` (edge_order 1), which the
    reference applies at mig_python.py:93.  Returns
    ``(uniform, h, ga, gb, gc)``: NumPy switches to the plain central
    difference only when every spacing is bit-identical."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim != 1 or x.shape[0] < 2:
        raise ValueError('Shape of array too small to calculate a numerical gradient, '
                         'at least (edge_order + 1) elements are required.')
    diffx = np.diff(x)
    if (diffx == diffx[0]).all():
        return True, float(diffx[0]), None, None, None
    n = x.shape[0]
    ga = np.zeros(n)
    gb = np.zeros(n)
    gc = np.zeros(n)
    dx1 = diffx[:-1]
    dx2 = diffx[1:]
    ga[1:-1] = -(dx2) / (dx1 * (dx1 + dx2))
    gb[1:-1] = (dx2 - dx1) / (dx1 * dx2)
    gc[1:-1] = dx1 / (dx2 * (dx1 + dx2))
    ga[0] = diffx[0]
    ga[-1] = diffx[-1]
    return False, 1.0, ga, gb, gc


def _kx(dat):
    """Horizontal wavenumbers, mig_python.py:163-168 / :262-267."""
    if np.mean(dat.trace_int) <= 0:
        Warning("The trace spacing, variable 'dat.trace_int', should be greater than 0. "
                "Using gradient(dat.dist) instead.")
        trace_int = np.gradient(dat.dist)
    else:
        trace_int = dat.trace_int
    return 2. * np.pi * np.fft.fftfreq(dat.tnum, d=np.mean(trace_int))


def _emit_metrics(dat, seconds, ngpus=1, ctx=None):
    """SURVEY section 5 "metrics": one JSON line per migration beside the reference's 'complete in N seconds' print
    (mig_python.py:121-122,206-207,285-286), on stderr, when ``$IMPDAR_METRICS`` is set: what ran (entry point, kernel,
    kernel and device milliseconds from HIP events: ``impdar_ctx_last_metrics``), the sizes, traces per second of the
    whole call and the device count."""
    if not os.environ.get('IMPDAR_METRICS'):
        return
    import json
    import sys
    rec = {}
    if ctx is not None:
        buf = C.create_string_buffer(1024)
        if _hip.load().impdar_ctx_last_metrics(ctx, buf, len(buf)) == 0:
            rec = json.loads(buf.value.decode())
    data = dat.__dict__.get('data', None)          # (a resident radargram has no host array: do not fetch it for this)
    rec.update(impdar_metrics=1, snum=int(dat.snum), tnum=int(dat.tnum),
               dtype=str(data.dtype) if hasattr(data, 'dtype') else 'resident',
               wall_s=round(float(seconds), 6), traces_per_s=round(float(dat.tnum) / max(float(seconds), 1e-9), 3),
               devices=int(ngpus))
    sys.stderr.write(json.dumps(rec) + '\n')
    sys.stderr.flush()


# ---------------------------------------------------------------------------
# Kirchhoff
# ---------------------------------------------------------------------------
def migrationKirchhoff(dat, vel=1.69e8, nearfield=False, mode=None, ngpus=None):
    """Kirchhoff diffraction-summation migration (mig_python.py:63-123).

    ``ngpus`` (extension; default ``$IMPDAR_NGPUS``): above 1 the radargram is sharded by output-trace blocks
    over that many GPUs of the node, one worker process per GPU (``impdar_amd.parallel.run_sharded``).

    ``mode`` (extension): None/'auto' picks the fp32 LDS-ring kernel for
    float32 data on uniform grids and the fp64 reference-order kernel
    otherwise; 'exact' / 'fast' force one ($IMPDAR_KIRCH_MODE overrides None);
    'fast' on float64 or integer data converts it to float32 first.
    Output is float64 like the reference.
    """
    print('Kirchhoff Migration (diffraction summation) of %.0fx%.0f matrix' % (dat.snum, dat.tnum))
    print('Using the MI355X HIP engine')
    _check_data_shape(dat)
    start = time.time()
    if mode is None:
        mode = os.environ.get('IMPDAR_KIRCH_MODE') or None
    if mode not in _MODES:
        raise ValueError('mode must be one of auto, exact, fast')
    src = np.asarray(dat.data)
    if mode == 'fast' and src.dtype != np.float32:
        # explicit opt-in to the float32 kernel for float64 / integer data
        src = src.astype(np.float32)
    from ... import parallel
    ngpus = parallel.ngpus_requested() if ngpus is None else int(ngpus)
    if ngpus > 1:
        # before this process touches the GPU: the ranks are child processes, one per device
        if np.shape(dat.dist) != (dat.tnum,):
            raise ValueError('dist must have one entry per trace')
        dat.data = parallel.run_sharded(src, dat.dist, dat.travel_time, vel=vel, nearfield=nearfield, ngpus=ngpus,
                                        mode=mode or 'auto')
        print('')
        print('Kirchhoff Migration of %.0fx%.0f matrix complete in %.2f seconds on %d GPUs'
              % (dat.snum, dat.tnum, time.time() - start, ngpus))
        _emit_metrics(dat, time.time() - start, ngpus)
        return dat
    lib = _hip.load()
    ctx = _hip.context()
    data, code = _device_data(src)
    tt_sec = np.ascontiguousarray(dat.travel_time / 1.0e6, dtype=np.float64)
    uniform, h, ga, gb, gc = gradient_coefficients(tt_sec)
    dist = np.ascontiguousarray(dat.dist, dtype=np.float64) * 1.0e3
    if dist.shape != (dat.tnum,):
        raise ValueError('dist must have one entry per trace')
    out = np.empty((dat.snum, dat.tnum), dtype=np.float64)
    _, p_dist = _hip.as_dp(dist)
    _, p_tt = _hip.as_dp(tt_sec)
    ka, p_ga = _hip.as_dp(ga)
    kb, p_gb = _hip.as_dp(gb)
    kc, p_gc = _hip.as_dp(gc)
    rc = lib.impdar_kirchhoff(ctx, data.ctypes.data_as(C.c_void_p), code, dat.snum, dat.tnum, p_dist, p_tt,
                              float(vel), int(bool(nearfield)), int(uniform), h, p_ga, p_gb, p_gc,
                              _MODES[mode], out.ctypes.data_as(_hip._dp))
    _hip.check(rc, 'impdar_kirchhoff')
    dat.data = out
    print('')
    print('Kirchhoff Migration of %.0fx%.0f matrix complete in %.2f seconds'
          % (dat.snum, dat.tnum, time.time() - start))
    _emit_metrics(dat, time.time() - start, 1, ctx)
    return dat


# ---------------------------------------------------------------------------
# Stolt
# ---------------------------------------------------------------------------
def migrationStolt(dat, vel=1.68e8, htaper=100, vtaper=1000):
    """Stolt f-k migration (mig_python.py:126-208): taper, rfft2 over
    (time, trace), Stolt stretch with linear interpolation along omega,
    obliquity scaling, inverse transform.  float32 data stays float32 (the
    reference's complex64 path under NumPy >= 2); everything else float64.
    Integer data is truncated back to its dtype after the taper (:157)."""
    print('Stolt Migration (f-k migration) of %.0fx%.0f matrix' % (dat.snum, dat.tnum))
    _check_data_shape(dat)
    start = time.time()
    lib = _hip.load()
    ctx = _hip.context()
    src = np.asarray(dat.data)
    pre_tapered = False
    if not np.issubdtype(src.dtype, np.floating):
        # integer dtypes: the taper product is truncated to the integer type
        # before the transform (:157); do that cast on the host so the device
        # sees exactly the reference's post-taper values
        it = np.arange(dat.tnum)
        ks = np.arange(dat.snum)
        hh = np.minimum(it, it[::-1]) / htaper
        vv = np.minimum(ks, ks[::-1]) / vtaper
        hh[hh > 1.] = 1.
        vv[vv > 1.] = 1.
        src = (src * hh[None, :] * vv[:, None]).astype(src.dtype)
        pre_tapered = True
    data, code = _device_data(src)
    ws = 2. * np.pi * np.fft.rfftfreq(dat.snum, d=dat.dt)
    kx = _kx(dat)
    print(kx.shape, ws.shape, (dat.snum // 2 + 1, dat.tnum))
    nout = 2 * (dat.snum // 2)
    out = np.empty((nout, dat.tnum), dtype=data.dtype)
    _, p_kx = _hip.as_dp(kx)
    _, p_ws = _hip.as_dp(ws)
    # NaN taper lengths tell the device the taper has already been applied
    ht = float('nan') if pre_tapered else float(htaper)
    vt = float('nan') if pre_tapered else float(vtaper)
    rc = lib.impdar_stolt(ctx, data.ctypes.data_as(C.c_void_p), code, dat.snum, dat.tnum, p_kx, p_ws,
                          float(vel), ht, vt, out.ctypes.data_as(C.c_void_p))
    _hip.check(rc, 'impdar_stolt')
    dat.data = out
    print('')
    print('Stolt Migration of %.0fx%.0f matrix complete in %.2f seconds'
          % (dat.snum, dat.tnum, time.time() - start))
    _emit_metrics(dat, time.time() - start, 1, ctx)
    return dat


# ---------------------------------------------------------------------------
# phase shift
# ---------------------------------------------------------------------------
def migrationPhaseShift(dat, vel=1.69e8, vel_fn=None, htaper=100, vtaper=1000, ngpus=None, **genfromtxt_kwargs):
    """Phase-shift (Gazdag) migration (mig_python.py:211-287; kernel semantics
    :361-493).  Constant ``vel``, a 2-column (v, z) table or a 3-column
    (v, z, x) table / ``vel_fn`` file; the last runs the 2-D v(x, z) Fourier
    finite-difference branch (:428-432, 448-487, 496-540) in float64.

    ``ngpus`` (extension; default ``$IMPDAR_NGPUS``): above 1 the constant-velocity and v(z) forms are sharded
    over the wavenumbers on that many GPUs (``impdar_amd.parallel.run_sharded_phaseshift``)."""
    return _phase_shift(dat, vel, vel_fn, htaper, vtaper, genfromtxt_kwargs, None, ngpus)


def _phase_shift(dat, vel, vel_fn, htaper, vtaper, genfromtxt_kwargs, dev, ngpus=None):
    """``dev``: None -> ``dat.data`` (host); a resident DeviceArray -> migrated where it is, the result
    replaces ``dat._dev`` (constant v and 1-D v(z); the 2-D branch goes through the host)."""
    print('Phase-Shift Migration of %.0fx%.0f matrix' % (dat.snum, dat.tnum))
    if dev is None:
        _check_data_shape(dat)
        src_dtype = np.asarray(dat.data).dtype
    else:
        if dev.shape != (dat.snum, dat.tnum):
            raise ValueError('The input array must be of size (snum, tnum)')
        src_dtype = dev.dtype
    start = time.time()
    if not np.issubdtype(src_dtype, np.floating):
        # the reference's in-place ``dat.data *= H*V`` (:258) cannot cast float -> int
        raise TypeError("Cannot cast ufunc 'multiply' output from dtype('float64') to dtype('%s') "
                        "with casting rule 'same_kind'" % src_dtype)
    from ... import parallel
    ngpus = (parallel.ngpus_requested() if ngpus is None else int(ngpus)) if dev is None else 0
    lib = None if ngpus > 1 else _hip.load()
    # with ngpus > 1 the ranks are child processes, one per device: this process stays off the GPU until it is
    # known that the run is not sharded (the 2-D branch)
    ctx = None if ngpus > 1 else (_hip.context() if dev is None else dev.ctx)
    if dev is None:
        data, code = _device_data(dat.data)
    else:
        data, code = None, _hip.dtype_code(dev.dtype)
    nt = int(2 ** (np.ceil(np.log(dat.snum) / np.log(2))))
    kx = _kx(dat)
    ws = 2. * np.pi * np.fft.fftfreq(nt, d=dat.dt)
    if vel_fn is not None:
        try:
            vel = np.genfromtxt(vel_fn, **genfromtxt_kwargs)
            print('Velocities loaded from %s.' % vel_fn)
        except Exception:
            raise TypeError('File %s was given for input velocity array, but cannot be loaded. '
                            'Please reformat to txt file.' % vel_fn)
    vmig = getVelocityProfile(dat, vel)
    if not hasattr(vmig, '__len__'):
        print('Constant velocity %s m/usec' % (vmig / 1e6))
        vconst, vm, vlen, p_vm = float(vmig), None, 0, None
    else:
        if not hasattr(vmig, 'shape'):
            raise ValueError('vmig needs to be an array or float')
        if len(vmig) != dat.snum:
            raise ValueError('Interpolated velocity profile is not the length of the number of samples in a trace.')
        if hasattr(vmig[0], '__len__'):
            print('2-D velocity structure, Fourier Finite-Difference Migration')
            vm2 = np.ascontiguousarray(vmig, dtype=np.float64)
            if vm2.shape != (dat.snum, dat.tnum):
                raise ValueError('2-D velocity array must have shape (snum, tnum)')
            if dev is not None:
                dat.from_device()
            lib = _hip.load()
            ctx = _hip.context() if ctx is None else ctx
            d64 = np.ascontiguousarray(dat.data, dtype=np.float64)
            out = np.empty((dat.snum, dat.tnum), dtype=np.float64)
            tt_us, p_tt = _hip.as_dp(dat.travel_time)
            _, p_kx = _hip.as_dp(kx)
            _, p_ws = _hip.as_dp(ws)
            rc = lib.impdar_phaseshift_ffd(ctx, _hip.as_dp(d64)[1], dat.snum, dat.tnum, nt, p_kx, p_ws, float(dat.dt),
                                           p_tt, _hip.as_dp(vm2)[1], float(np.mean(dat.trace_int)), float(htaper),
                                           float(vtaper), _hip.as_dp(out)[1])
            _hip.check(rc, 'impdar_phaseshift_ffd')
            dat.data = out
            if dev is not None:
                dat.to_device()
            print('')
            print('Phase-Shift Migration of %.0fx%.0f matrix complete in %.2f seconds'
                  % (dat.snum, dat.tnum, time.time() - start))
            _emit_metrics(dat, time.time() - start, 1, ctx)
            return dat
        print('1-D velocity structure, Gazdag Migration')
        vconst = 0.0
        vm, p_vm = _hip.as_dp(vmig)
        vlen = dat.snum
    if ngpus > 1:
        dat.data = parallel.run_sharded_phaseshift(data, nt, kx, ws, float(dat.dt), dat.travel_time, vconst, vm,
                                                   htaper, vtaper, ngpus=ngpus)
        print('')
        print('Phase-Shift Migration of %.0fx%.0f matrix complete in %.2f seconds on %d GPUs'
              % (dat.snum, dat.tnum, time.time() - start, ngpus))
        _emit_metrics(dat, time.time() - start, ngpus)
        return dat
    tt_us, p_tt = _hip.as_dp(dat.travel_time)
    _, p_kx = _hip.as_dp(kx)
    _, p_ws = _hip.as_dp(ws)
    if dev is None:
        # upload, migrate on the device, and bring the result back as float64 (the reference returns
        # ifft(...).real, :282) through the threaded widening download
        d_in = _hip.DeviceArray.from_host(ctx, data)
        d_out = _hip.DeviceArray(ctx, d_in.shape, d_in.dtype)
        try:
            rc = lib.impdar_phaseshift_dev(ctx, d_in.ptr, code, dat.snum, dat.tnum, nt, p_kx, p_ws, float(dat.dt), p_tt,
                                           vconst, p_vm, vlen, float(htaper), float(vtaper), d_out.ptr)
            _hip.check(rc, 'impdar_phaseshift')
            dat.data = d_out.to_host_f64()
        finally:
            d_in.free()
            d_out.free()
    else:
        d_out = _hip.DeviceArray(ctx, dev.shape, dev.dtype)
        rc = lib.impdar_phaseshift_dev(ctx, dev.ptr, code, dat.snum, dat.tnum, nt, p_kx, p_ws, float(dat.dt), p_tt,
                                       vconst, p_vm, vlen, float(htaper), float(vtaper), d_out.ptr)
        if rc:
            d_out.free()
        _hip.check(rc, 'impdar_phaseshift')
        _hip.check(lib.impdar_ctx_sync(ctx), 'impdar_ctx_sync')
        dev.free()
        dat._dev = d_out
        dat._dev_widen = True          # float64 on the way back, as the host path
    print('')
    print('Phase-Shift Migration of %.0fx%.0f matrix complete in %.2f seconds'
          % (dat.snum, dat.tnum, time.time() - start))
    _emit_metrics(dat, time.time() - start, 1, ctx)
    return dat


def migrationTimeWavenumber(dat, vel=1.69e8, vel_fn=None, htaper=100, vtaper=1000):
    """The reference's T-K migration is a stub that only tapers the data in
    place (mig_python.py:290-355, loop body ``continue`` at :346-347); this
    reproduces exactly that, on the device."""
    print('Time-Wavenumber Migration of %.0fx%.0f matrix' % (dat.snum, dat.tnum))
    _check_data_shape(dat)
    start = time.time()
    if not np.issubdtype(np.asarray(dat.data).dtype, np.floating):
        raise TypeError("Cannot cast ufunc 'multiply' output from dtype('float64') to dtype('%s') "
                        "with casting rule 'same_kind'" % np.asarray(dat.data).dtype)
    lib = _hip.load()
    ctx = _hip.context()
    data = np.array(dat.data, order='C', copy=True)
    if data.dtype not in (np.float32, np.float64):
        data = data.astype(np.float64)
    rc = lib.impdar_taper(ctx, data.ctypes.data_as(C.c_void_p), _hip.dtype_code(data.dtype), dat.snum, dat.tnum,
                          float(htaper), float(vtaper))
    _hip.check(rc, 'impdar_taper')
    dat.data = data
    _kx(dat)
    print('')
    print('Time-Wavenumber Migration of %.0fx%.0f matrix complete in %.2f seconds'
          % (dat.snum, dat.tnum, time.time() - start))
    _emit_metrics(dat, time.time() - start, 1, ctx)
    return dat


# ---------------------------------------------------------------------------
# velocity profile (host, O(snum))
# ---------------------------------------------------------------------------
def _interp_checked(x, y, xnew):
    """Linear interpolation with scipy.interpolate.interp1d's default
    contract: abscissae sorted first, ValueError outside their range."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    order = np.argsort(x, kind='mergesort')
    x, y = x[order], y[order]
    xnew = np.asarray(xnew, dtype=np.float64)
    if np.any(xnew < x[0]):
        raise ValueError('A value in x_new is below the interpolation range.')
    if np.any(xnew > x[-1]):
        raise ValueError('A value in x_new is above the interpolation range.')
    # numpy.interp is what SciPy's 1-D linear interp1d evaluates with (same slope formula; repeated
    # abscissae resolve to the last of the equal knots instead of 0/0)
    return np.interp(xnew, x, y)


# Error messages of the reference's velocity-table checks (mig_python.py:572-573, :581, :587-588, :614, :632, :639):
# same exception type and text, raised from one table instead of scattered branches.
_VEL_ERRORS = {
    'shape': 'If non-constant vel, inputs needs to be 2d (v, z) or (v, z, x)',
    'one_layer': 'It does not make sense to only give one layer of velocity--'
                 'if you want constant velocity just input v',
    'coverage': 'Your velocity data doesnt come close to covering the depths in the data',
    'no_dist': 'The distance vector was never set.',
    'range': 'Two-way travel time array extends outside of interpolation range',
    'columns': 'Input must be 2d with 2 or 3 columns',
}


def _vel_fail(key):
    raise ValueError(_VEL_ERRORS[key])


def _velocity_of_time(depth, time_of_depth, twtt, check_end=False):
    """Migration velocity on the sample times from a travel-time curve t(z) given on ``depth``: invert it to z(t) on
    ``twtt`` by linear interpolation and differentiate, v = 2 dz/dt (mig_python.py:600-604, :628-636).  Shared by
    the layered and the lateral branch."""
    if check_end and twtt[-1] > time_of_depth[-1]:
        _vel_fail('range')
    z_of_t = _interp_checked(time_of_depth, depth, twtt)
    return 2. * np.gradient(z_of_t, twtt)


def _layered_profile(vel_v, vel_z, twtt):
    """2-column (v, z) table -> v(t) (mig_python.py:582-604)."""
    depth = np.max(vel_v) / 2. * twtt                 # depth axis for the largest possible penetration
    depth[0] = twtt[0] * vel_v[0] / 2.
    shallow, deep = np.nanmin(depth), np.nanmax(depth)
    starts_late = vel_z[0] > 1.1 * shallow and vel_z[0] / deep > 1.0e-3
    ends_early = vel_z[-1] * 1.1 < deep
    if starts_late or ends_early:
        _vel_fail('coverage')
    # a table that stops just short of the depth range is stretched to it with its end velocities
    if vel_z[0] > shallow:
        vel_v, vel_z = np.insert(vel_v, 0, vel_v[np.argmin(vel_z)]), np.insert(vel_z, 0, shallow)
    if vel_z[-1] < deep:
        vel_v, vel_z = np.append(vel_v, vel_v[np.argmax(vel_z)]), np.append(vel_z, deep)
    table_t = 2. * vel_z / vel_v                      # local two-way time of each table row (not integrated)
    return _velocity_of_time(depth, _interp_checked(vel_z, table_t, depth), twtt)


def _lateral_profile(vel_v, vel_z, vel_x, twtt, dat):
    """3-column (v, z, x) table -> v(t, trace) (mig_python.py:607-636): nearest table row for every (trace, depth)
    node, then the integrated travel time down each trace."""
    if dat.dist is None or np.all(np.asarray(dat.dist) == 0):
        _vel_fail('no_dist')
    depth = np.linspace(np.min(vel_v) * twtt[0], np.max(vel_v) * twtt[-1], dat.snum) / 2.
    along = np.asarray(dat.dist, dtype=np.float64)
    thickness = np.diff(depth)
    out = np.zeros((dat.snum, dat.tnum))
    for i in range(dat.tnum):
        nearest = np.argmin((along[i] - vel_x[None, :]) ** 2 + (depth[:, None] - vel_z[None, :]) ** 2, axis=1)
        slowness = 1. / vel_v[nearest]
        # the reference integrates 1/v over the samples ABOVE each depth (trapz over the first j points, :626):
        # a cumulative trapezoid shifted by one sample
        layers = np.cumsum(thickness * (slowness[1:] + slowness[:-1]) / 2.)
        table_t = 2. * np.concatenate([[0., 0.], layers[:-1]])
        out[:, i] = _velocity_of_time(depth, _interp_checked(depth, table_t, depth), twtt, check_end=True)
    return out


def getVelocityProfile(dat, vels_in):
    """Map a velocity table onto the samples of the traces (mig_python.py:543-643).  Scalar -> returned unchanged;
    2-column (v, z) -> 1-D profile of length snum; 3-column (v, z, x) -> (snum, tnum) array; malformed tables raise
    the reference's ValueErrors (``_VEL_ERRORS``)."""
    if not hasattr(vels_in, '__len__'):
        return vels_in
    start = time.time()
    print('Interpolating the velocity profile.')
    shape = np.shape(vels_in)
    if len(shape) != 2 or shape[1] == 1:
        _vel_fail('shape')
    if shape[0] == 1:
        _vel_fail('one_layer')
    if shape[1] not in (2, 3):
        _vel_fail('columns')
    table = np.asarray(vels_in, dtype=np.float64)
    twtt = np.asarray(dat.travel_time, dtype=np.float64).copy() / 1.0e6
    if shape[1] == 2:
        vmig = _layered_profile(table[:, 0], table[:, 1], twtt)
    else:
        vmig = _lateral_profile(table[:, 0], table[:, 1], table[:, 2], twtt, dat)
    print('Velocity profile finished in %.2f seconds.' % (time.time() - start))
    return vmig

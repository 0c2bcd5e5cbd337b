"""CPU oracle for the ImpDAR migration hot path.  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the *behaviour* of the reference's
``impdar.lib.migrationlib.mig_python`` (cited per function as
``mig_python.py:<line>``).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product
(``impdar_amd``) never does and has no CPU fallback.

Parity pinning: the reference's own tests hold no numerical vector for any
migration (they run on all-zeros input), so this oracle is pinned against
outputs of the reference itself, generated in the build container by
``tests/golden/make_golden.py`` (which imports ``/root/reference/src``) and
committed as ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks
every function here against those vectors.

Conventions (reference ``RadarData/__init__.py:132-174``): ``data`` has shape
``(snum, tnum)``; ``travel_time`` is in microseconds; ``dist`` in km;
``dt`` in seconds; ``trace_int`` in metres.
"""
import numpy as np

__all__ = [
    "check_data_shape", "taper", "kirchhoff", "kirchhoff_literal",
    "kirchhoff_pick", "stolt", "phase_shift", "phase_shift_tk",
    "time_wavenumber", "get_velocity_profile", "count_pairs", "phase_shift_ffd_tk",
]


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def check_data_shape(data, snum, tnum):
    """mig_python.py:646-648 -- ValueError unless data.shape == (snum, tnum)."""
    if np.size(data, 1) != tnum or np.size(data, 0) != snum:
        raise ValueError('The input array must be of size (snum, tnum)')


def taper(snum, tnum, htaper, vtaper):
    """Linear edge taper weights (mig_python.py:152-156, 253-257, 330-334).

    Returns (h, v): h[i] = min(i, tnum-1-i)/htaper clipped to 1 (per trace),
    v[k] = min(k, snum-1-k)/vtaper clipped to 1 (per sample).
    """
    it = np.arange(tnum)
    ks = np.arange(snum)
    h = np.minimum(it, it[::-1]) / htaper
    v = np.minimum(ks, ks[::-1]) / vtaper
    h = np.where(h > 1., 1., h)
    v = np.where(v > 1., 1., v)
    return h, v


def _apply_taper(data, htaper, vtaper, inplace_form=False):
    """Taper product.  Stolt computes ``(data*H)*V`` (mig_python.py:157); the
    phase-shift and T-K routines compute ``data *= (H*V)`` (:258, :335)."""
    snum, tnum = data.shape
    h, v = taper(snum, tnum, htaper, vtaper)
    if inplace_form:
        return data * (h[None, :] * v[:, None])
    return (data * h[None, :]) * v[:, None]


def _kx(tnum, trace_int, dist):
    """Horizontal wavenumbers (mig_python.py:163-168, 262-267)."""
    if np.mean(trace_int) <= 0:
        ti = np.gradient(dist)          # km, as the reference does
    else:
        ti = trace_int
    return 2. * np.pi * np.fft.fftfreq(tnum, d=np.mean(ti))


# --------------------------------------------------------------------------
# Kirchhoff
# --------------------------------------------------------------------------
def kirchhoff_pick(t, tt_sec):
    """Nearest-sample pick = argmin_k |tt[k]-t| with ties to the lower k.

    Closed form of mig_python.py:49 for monotonically increasing ``tt_sec``.
    """
    snum = len(tt_sec)
    k0 = np.searchsorted(tt_sec, t, side='right') - 1
    k0 = np.clip(k0, 0, snum - 1)
    k1 = np.minimum(k0 + 1, snum - 1)
    up = np.abs(tt_sec[k1] - t) < np.abs(tt_sec[k0] - t)
    return np.where(up, k1, k0)


def kirchhoff_loop(gradD, dist_m, zs, zs2, tt_sec, vel, max_travel_time, data=None, nearfield=False, traces=None):
    """migrationKirchhoffLoop (mig_python.py:35-60) with the caller's own tables, as the reference's native hook
    receives them (mig_cython.h:11): ``zs`` / ``zs2`` depth and depth squared per sample, ``max_travel_time`` the
    limit beyond which a pair is dropped (:52).  O(in-aperture pairs): the argmin of :49 in closed form."""
    gradD = np.ascontiguousarray(gradD, dtype=np.float64)
    snum, tnum = gradD.shape
    tt = np.asarray(tt_sec, dtype=np.float64)
    dist = np.asarray(dist_m, dtype=np.float64)
    zs = np.asarray(zs, dtype=np.float64)
    zs2 = np.asarray(zs2, dtype=np.float64)
    d64 = None if data is None else np.asarray(data, dtype=np.float64)
    out = np.zeros((snum, tnum), dtype=np.float64)
    cols = np.arange(tnum)
    todo = range(tnum) if traces is None else traces
    for xi in todo:
        dx2 = (dist - dist[xi]) ** 2.
        rs = np.sqrt(dx2[None, :] + zs2[:, None])            # (snum, tnum), :44
        with np.errstate(invalid='ignore', divide='ignore'):
            cost = zs[:, None] / rs                          # :46-47
        t = 2. * rs / vel
        k = kirchhoff_pick(t, tt)                            # :49
        g = gradD[k, cols[None, :]]
        g = np.where(t > max_travel_time, 0., g)             # :51-52
        acc = np.nansum(g * cost / vel, axis=1)              # :53
        if nearfield:
            d = d64[k, cols[None, :]]
            d = np.where(t > max_travel_time, 0., d)
            with np.errstate(invalid='ignore', divide='ignore'):
                acc = acc + np.nansum(d * cost / rs ** 2., axis=1)      # :55-58
        out[:, xi] = acc / (2. * np.pi)                      # :60
    return out


def kirchhoff(data, travel_time_us, dist_km, vel=1.69e8, nearfield=False,
              traces=None):
    """Diffraction-sum migration, O(in-aperture pairs).

    Restates mig_python.py:63-123 (driver) and :35-60 (loop).  Returns the
    float64 migrated array (snum, tnum).  ``traces`` optionally restricts the
    output traces computed (others left 0) -- used for bounded CPU timing.
    """
    data = np.asarray(data)
    tt = np.asarray(travel_time_us) / 1.0e6
    # :93 -- gradient keeps float32 for float32 input, float64 otherwise
    gradD = np.ascontiguousarray(np.gradient(data, tt, axis=0), dtype=np.float64)
    zs = vel * tt / 2.0                                      # :101
    zs2 = zs ** 2.                                           # :102
    dist = np.ascontiguousarray(dist_km, dtype=np.float64) * 1.0e3      # :108
    return kirchhoff_loop(gradD, dist, zs, zs2, tt, vel, np.max(tt), data.astype(np.float64), nearfield, traces)


def kirchhoff_literal(data, travel_time_us, dist_km, vel=1.69e8, nearfield=False):
    """Literal O((snum*tnum)^2) form: argmin over the whole |tt - t| matrix
    for every output sample (mig_python.py:39-60).  Small cases only."""
    data = np.asarray(data)
    snum, tnum = data.shape
    tt = np.asarray(travel_time_us) / 1.0e6
    gradD = np.asarray(np.gradient(data, tt, axis=0), dtype=np.float64)
    d64 = data.astype(np.float64)
    tmax = np.max(tt)
    zs = vel * tt / 2.0
    dist = np.asarray(dist_km, dtype=np.float64) * 1.0e3
    out = np.zeros((snum, tnum))
    cols = np.arange(tnum)
    for xi in range(tnum):
        for ti in range(snum):
            rs = np.sqrt((dist - dist[xi]) ** 2. + zs[ti] ** 2.)
            with np.errstate(invalid='ignore'):
                cost = zs[ti] / rs
            t = 2. * rs / vel
            k = np.argmin(np.abs(tt[:, None] - t[None, :]), axis=0)
            g = gradD[k, cols].copy()
            g[t > tmax] = 0.
            s = np.nansum(g * cost / vel)
            if nearfield:
                d = d64[k, cols].copy()
                d[t > tmax] = 0.
                with np.errstate(invalid='ignore', divide='ignore'):
                    s += np.nansum(d * cost / rs ** 2.)
            out[ti, xi] = s / (2. * np.pi)
    return out


def count_pairs(snum, tnum, dt, dx, vel, t0_us=0.0):
    """Number of in-aperture (output sample, input trace) pairs for a uniform
    geometry: #{(ti, xi, j): 2*sqrt((x_j-x_xi)^2+z_ti^2)/vel <= t_max}.
    This is the unit SURVEY 8(d) prices the Kirchhoff roofline in."""
    tt = t0_us * 1e-6 + np.arange(snum) * dt
    tmax = tt[-1]
    r2 = (vel * tmax / 2.) ** 2 - (vel * tt / 2.) ** 2       # allowed dx^2
    total = 0
    n = np.arange(tnum)
    for ti in range(snum):
        if r2[ti] < 0:
            continue
        h = int(np.floor(np.sqrt(r2[ti]) / dx + 1e-12))
        # traces j with |j - xi| <= h, clipped to the profile
        lo = np.maximum(n - h, 0)
        hi = np.minimum(n + h, tnum - 1)
        total += int(np.sum(hi - lo + 1))
    return total


# --------------------------------------------------------------------------
# Stolt
# --------------------------------------------------------------------------
def stolt(data, dt, trace_int, dist_km=None, vel=1.68e8, htaper=100, vtaper=1000):
    """Stolt f-k migration (mig_python.py:126-208).

    Returns the migrated array with 2*(snum//2) rows.  dtype rules follow the
    reference: taper result is cast back to the input dtype (:157), float32
    input runs a complex64 FFT path and yields float32.
    """
    data = np.asarray(data)
    snum, tnum = data.shape
    tap = _apply_taper(data, htaper, vtaper).astype(data.dtype)
    FK = np.fft.rfft2(tap, axes=(1, 0))                      # (snum//2+1, tnum)
    ws = 2. * np.pi * np.fft.rfftfreq(snum, d=dt)
    kx = _kx(tnum, trace_int, dist_km)
    nw = len(ws)
    KK = np.zeros_like(FK)
    nz = snum // 2
    kz = 2. * ws / vel
    # linear interpolation along omega within each kx column, clamped at the
    # last knot (FITPACK bispev with kx=ky=1 evaluated at a kx knot, :171-190)
    wsj = vel / 2. * np.sqrt(kz[:nz, None] ** 2. + kx[None, :] ** 2.)
    wsj = np.minimum(wsj, ws[-1])
    dw = ws[1] - ws[0] if nw > 1 else 1.0
    i0 = np.clip(np.floor(wsj / dw).astype(np.int64), 0, nw - 2)
    # guard against floor() landing one knot off because of rounding
    i0 = np.where(ws[i0] > wsj, i0 - 1, i0)
    i0 = np.where(ws[np.minimum(i0 + 1, nw - 1)] <= wsj, np.minimum(i0 + 1, nw - 2), i0)
    i0 = np.clip(i0, 0, nw - 2)
    w = (wsj - ws[i0]) / (ws[i0 + 1] - ws[i0])
    cols = np.arange(tnum)[None, :]
    KK[:nz] = (1. - w) * FK[i0, cols] + w * FK[i0 + 1, cols]
    with np.errstate(invalid='ignore', divide='ignore'):
        scaling = kz[:, None] / np.sqrt(kx[None, :] ** 2. + kz[:, None] ** 2.)
    KK = KK * scaling.astype(KK.real.dtype) if KK.dtype == np.complex64 else KK * scaling
    KK[0, 0] = 0. + 0j
    return np.fft.irfft2(KK, axes=(1, 0))


# --------------------------------------------------------------------------
# Phase shift (Gazdag)
# --------------------------------------------------------------------------
def _ffd_stencil_apply(x):
    """stencil * x for stencil = Sp_Matr(N, -2, 1, 1) (mig_python.py:528-540) as the
    reference actually builds it: the k3 = 0 ``setdiag(..., k=0)`` at :533 overwrites
    the main diagonal with zeros, row 0 is e_0 and the last row is all ones."""
    y = np.empty_like(x)
    y[1:-1] = x[:-2] + x[2:]
    y[0] = x[0]
    y[-1] = np.sum(x)
    return y


def phase_shift_ffd_tk(FK, vmig, kx, ws, dt, travel_time_us, trace_int_mean, snum, tnum,
                       alpha=0.5, beta=0.25):
    """2-D v(x,z) branch of phaseShift (mig_python.py:428-432,448-487) with
    fourierFiniteDiff (:496-525).  Literal loops (small cases only): the update of
    one frequency uses the previous frequency's field (``FFX_last`` is a single
    variable, :478), so the whole (tau, omega) nest is one serial chain."""
    FK = np.array(FK, dtype=np.complex128)
    vmig = np.asarray(vmig, dtype=np.float64)
    tt = np.asarray(travel_time_us, dtype=np.float64)
    TK = np.zeros((snum, len(kx)), dtype=np.complex128)
    dx = trace_int_mean
    FFX_last = 0.
    for itau in range(snum):
        tau = tt[itau] / 1.0e6
        vbg = np.min(vmig[itau])
        vfg = vmig[itau] - vbg
        ufg = 1. / vmig[itau] - 1. / vbg
        for iw in range(len(ws)):
            w = ws[iw]
            if w == 0.0:
                w = 1.0e-10 / dt
            coss = 1.0 + 0j - (0.5 * vbg * kx / w) ** 2.
            phase = (-w * dt * np.sqrt(coss)).real
            FK[iw] *= np.conj(np.cos(phase) + 1j * np.sin(phase))
            FFX = np.fft.ifft(FK[iw])
            phase2 = 2. * ufg * w * dt + 1. * vbg * w * dt
            FFX = FFX * (np.cos(phase2) + 1j * np.sin(phase2))
            if itau > 0:
                coeff1 = dt * alpha * vfg ** 2. / (1j * 4. * w * dx ** 2.)
                coeff2 = -beta * vfg ** 2. / (4. * w ** 2. * dx ** 2.)
                sf = _ffd_stencil_apply(FFX)
                sl = _ffd_stencil_apply(FFX_last)
                FFX = FFX_last + coeff1 * sf + coeff2 * (sf - sl)
            FFX_last = FFX
            FK[iw] = np.fft.fft(FFX)
            FK[iw, coss <= (tau / tt[-1] / 1e6) ** 2.] = 0.0 + 0j
            TK[itau] += FK[iw]
    TK = TK[:, :tnum]
    return TK / snum


def phase_shift_tk(FK, vmig, kx, ws, dt, travel_time_us, snum, tnum, trace_int_mean=None):
    """phaseShift (mig_python.py:361-493): FK (nt, tnum) -> TK (snum, tnum).

    ``vmig`` scalar -> constant-velocity branch (:396-420);
    1-D array of length snum -> Gazdag v(z) branch (:438-487);
    2-D (snum, tnum) -> Fourier finite-difference branch (phase_shift_ffd_tk).
    FK is not modified (the reference mutates it in the v(z) branch).
    """
    FK = np.array(FK, dtype=np.complex128)
    TK = np.zeros((snum, len(kx)), dtype=np.complex128)
    w = np.where(ws == 0.0, 1e-10 / dt, ws)                  # :400-402, :445-447
    if not hasattr(vmig, '__len__'):
        vkx2 = (vmig * kx / 2.) ** 2.
        prop = vkx2[None, :] < (w ** 2.)[:, None]
        with np.errstate(invalid='ignore'):
            root = np.sqrt(np.where(prop, 1.0 - vkx2[None, :] / (w ** 2.)[:, None], 0.))
        ph = w[:, None] * dt * root
        cp = np.cos(ph) + 1j * np.sin(ph)
        F = np.where(prop, FK, 0.)
        for itau in range(snum):
            F = F * cp
            TK[itau] = F.sum(axis=0)
    else:
        vmig = np.asarray(vmig)
        if len(vmig) != snum:
            raise ValueError('Interpolated velocity profile is not the length of the number of samples in a trace.')
        if vmig.ndim != 1:
            return phase_shift_ffd_tk(FK, vmig, kx, ws, dt, travel_time_us, trace_int_mean, snum, tnum)
        tt = np.asarray(travel_time_us, dtype=np.float64)
        F = FK
        for itau in range(snum):
            tau = tt[itau] / 1.0e6
            coss = 1.0 - (0.5 * vmig[itau] * kx[None, :] / w[:, None]) ** 2.
            ph = w[:, None] * dt * np.sqrt(np.maximum(coss, 0.))
            F = F * (np.cos(ph) + 1j * np.sin(ph))
            F = np.where(coss <= (tau / tt[-1] / 1e6) ** 2., 0., F)
            TK[itau] = F.sum(axis=0)
    TK = TK[:, :tnum]
    return TK / snum


def phase_shift(data, dt, trace_int, travel_time_us, dist_km=None, vel=1.69e8,
                htaper=100, vtaper=1000):
    """migrationPhaseShift (mig_python.py:211-287) for scalar, (v,z)-table or
    (v,z,x)-table ``vel``.  Returns float64 (snum, tnum)."""
    data = np.asarray(data)
    if not np.issubdtype(data.dtype, np.floating):
        # :258 in-place multiply of an integer array by floats raises
        raise TypeError('phase-shift migration needs floating-point data')
    snum, tnum = data.shape
    tap = _apply_taper(data, htaper, vtaper, inplace_form=True).astype(data.dtype)
    nt = int(2 ** np.ceil(np.log(snum) / np.log(2)))
    kx = _kx(tnum, trace_int, dist_km)
    ws = 2. * np.pi * np.fft.fftfreq(nt, d=dt)
    FK = np.fft.fft2(tap, (nt, tnum))
    vmig = get_velocity_profile(travel_time_us, vel, dist_km)
    TK = phase_shift_tk(FK, vmig, kx, ws, dt, travel_time_us, snum, tnum,
                        trace_int_mean=np.mean(trace_int))
    return np.fft.ifft(TK).real


def time_wavenumber(data, htaper=100, vtaper=1000):
    """migrationTimeWavenumber (mig_python.py:290-355) is a stub in the
    reference: it applies the taper in place and returns."""
    data = np.asarray(data)
    if not np.issubdtype(data.dtype, np.floating):
        raise TypeError('in-place taper of integer data raises in the reference')
    return _apply_taper(data, htaper, vtaper, inplace_form=True).astype(data.dtype)


# --------------------------------------------------------------------------
# velocity profile
# --------------------------------------------------------------------------
def _interp1d_strict(x, y, xnew):
    """scipy.interpolate.interp1d(x, y)(xnew) with its default behaviour:
    sort by x (stable), linear, ValueError outside [x.min, x.max]."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    order = np.argsort(x, kind='mergesort')
    x = x[order]
    y = y[order]
    xnew = np.asarray(xnew, dtype=np.float64)
    if np.any(xnew < x[0]):
        raise ValueError('A value in x_new is below the interpolation range.')
    if np.any(xnew > x[-1]):
        raise ValueError('A value in x_new is above the interpolation range.')
    # SciPy (1.15 here) evaluates a 1-D linear interp1d with numpy.interp: same slope formula as the
    # classic searchsorted rule, but repeated abscissae pick the LAST of the equal knots instead of
    # dividing by zero (this matters for the 3-column branch, whose t(z) starts with two zeros)
    return np.interp(xnew, x, y)


def _nearest_values(px, pz, pv, qx, qz):
    """griddata(..., method='nearest') (mig_python.py:617): value of the table
    point closest (Euclidean, in the raw (x, z) coordinates) to each query."""
    d2 = (qx[:, None] - px[None, :]) ** 2 + (qz[:, None] - pz[None, :]) ** 2
    return pv[np.argmin(d2, axis=1)]


def get_velocity_profile(travel_time_us, vels_in, dist_km=None):
    """getVelocityProfile (mig_python.py:543-643): scalar, 2-column (v, z) and
    3-column (v, z, x) tables (the last needs ``dist_km``; returns (snum, tnum))."""
    if not hasattr(vels_in, '__len__'):
        return vels_in
    if len(np.shape(vels_in)) != 2 or np.shape(vels_in)[1] == 1:
        raise ValueError('If non-constant vel, inputs needs to be 2d (v, z) or (v, z, x)')
    nlay, dimension = np.shape(vels_in)
    vels_in = np.asarray(vels_in, dtype=np.float64)
    vel_v = vels_in[:, 0]
    vel_z = vels_in[:, 1]
    twtt = np.asarray(travel_time_us, dtype=np.float64).copy() / 1.0e6
    if nlay == 1:
        raise ValueError('It does not make sense to only give one layer of velocity--if you want constant velocity just input v')
    if dimension == 2:
        zs = np.max(vel_v) / 2. * twtt
        zs[0] = twtt[0] * vel_v[0] / 2.
        zmin, zmax = np.nanmin(zs), np.nanmax(zs)
        if (vel_z[0] > 1.1 * zmin and vel_z[0] / zmax > 1.0e-3) or vel_z[-1] * 1.1 < zmax:
            raise ValueError('Your velocity data doesnt come close to covering the depths in the data')
        if vel_z[0] > zmin:
            vel_v = np.insert(vel_v, 0, vel_v[np.argmin(vel_z)])
            vel_z = np.insert(vel_z, 0, zmin)
        if vel_z[-1] < zmax:
            vel_v = np.append(vel_v, vel_v[np.argmax(vel_z)])
            vel_z = np.append(vel_z, zmax)
        vel_t = 2. * vel_z / vel_v
        tofz = _interp1d_strict(vel_z, vel_t, zs)
        zoft = _interp1d_strict(tofz, zs, twtt)
        return 2. * np.gradient(zoft, twtt)
    if dimension == 3:
        vel_x = vels_in[:, 2]
        snum = len(twtt)
        zs = np.linspace(np.min(vel_v) * twtt[0], np.max(vel_v) * twtt[-1], snum) / 2.       # :610-612
        if dist_km is None or np.all(np.asarray(dist_km) == 0):
            raise ValueError('The distance vector was never set.')                           # :614-615
        dist = np.asarray(dist_km, dtype=np.float64)
        tnum = len(dist)
        # nearest-neighbour gridding onto the (dist, zs) mesh (:616-618); note the table's x is
        # compared with dat.dist as is (the reference mixes km and m here)
        XS, ZS = np.meshgrid(dist, zs)
        VS = _nearest_values(vel_x, vel_z, vel_v, XS.ravel(), ZS.ravel()).reshape(XS.shape)
        vmig = np.zeros_like(VS)
        for i in range(tnum):                                                                # :622-636
            v = VS[:, i]
            # vel_t[j] = 2 * trapz(1/v[:j], zs[:j]): integral over the first j points
            seg = np.diff(zs) * (1. / v[1:] + 1. / v[:-1]) / 2.
            cum = np.concatenate([[0.], np.cumsum(seg)])          # cum[m] = integral over points 0..m
            vel_t = 2. * np.concatenate([[0.], cum[:-1]])         # j points -> cum[j-1]; j = 0 -> 0
            tofz = _interp1d_strict(zs, vel_t, zs)
            if twtt[-1] > tofz[-1]:
                raise ValueError('Two-way travel time array extends outside of interpolation range')
            zoft = _interp1d_strict(tofz, zs, twtt)
            vmig[:, i] = 2. * np.gradient(zoft, twtt)
        return vmig
    raise ValueError('Input must be 2d with 2 or 3 columns')

"""Synthetic radargrams used by the tests, the golden-vector generator and
``bench.py``.  RNG-free by default so every box builds
the same input.

Geometry follows the gprMax loader convention of the reference
(``src/impdar/lib/load/load_gprMax.py:56-66``): ``travel_time`` starts at 0 and
is in microseconds, ``dist`` is in km, ``trace_int`` in metres.
"""
import numpy as np

_PHI = 0.6180339887
_SQ2 = 0.4142135623


def geometry(snum, tnum, dt=1.0e-8, dx=1.0, t0_us=0.0):
    """Return dict(travel_time[us], dist[km], trace_int[m], dt[s])."""
    travel_time = t0_us + np.arange(snum) * dt * 1.0e6
    dist = np.arange(tnum) * dx / 1.0e3
    trace_int = np.ones(tnum) * dx
    return dict(travel_time=travel_time, dist=dist, trace_int=trace_int, dt=dt)


def diffractor_radargram(snum, tnum, vel=1.69e8, dt=1.0e-8, dx=1.0, fc=5.0e6,
                         ndiff=64, dtype=np.float64, t0_us=0.0, trace_lo=0,
                         trace_hi=None, chunk=512, threads=1):
    """``ndiff`` point diffractors imaged with a Ricker wavelet of centre
    frequency ``fc``: data[k, j] = sum_m A_m ricker(t_k - 2 r_mj / vel).

    ``trace_lo:trace_hi`` selects a column block of the full ``tnum``-trace
    radargram (used by multi-GPU ranks to build only their own shard).
    ``threads`` > 1 builds column chunks on a thread pool (NumPy releases the
    GIL inside the ufuncs); the values do not depend on it.
    """
    if trace_hi is None:
        trace_hi = tnum
    t = (t0_us * 1e-6 + np.arange(snum) * dt)
    tmax = t[-1]
    R = vel * tmax / 2.0
    m = np.arange(ndiff)
    xm = np.mod(m * _PHI, 1.0) * (tnum - 1) * dx
    zm = (0.05 + 0.9 * np.mod(m * _SQ2, 1.0)) * R
    out = np.empty((snum, trace_hi - trace_lo), dtype=dtype)
    a = (np.pi * fc) ** 2
    def build(c0):
        c1 = min(c0 + chunk, trace_hi)
        xj = np.arange(c0, c1) * dx
        acc = np.zeros((snum, c1 - c0), dtype=np.float64)
        for i in range(ndiff):
            r = np.sqrt((xj - xm[i]) ** 2 + zm[i] ** 2)
            amp = 1.0 / (1.0 + r / 100.0)
            u = t[:, None] - (2.0 * r / vel)[None, :]
            au2 = a * u * u
            acc += amp[None, :] * (1.0 - 2.0 * au2) * np.exp(-au2)
        out[:, c0 - trace_lo:c1 - trace_lo] = acc.astype(dtype)

    starts = list(range(trace_lo, trace_hi, chunk))
    if threads > 1 and len(starts) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(threads, len(starts))) as pool:
            list(pool.map(build, starts))
    else:
        for c0 in starts:
            build(c0)
    return out


def noise_radargram(snum, tnum, seed=0, dtype=np.float64):
    """White-noise stress variant (what the SURVEY section 6 timings used)."""
    return np.random.default_rng(seed).standard_normal((snum, tnum)).astype(dtype)

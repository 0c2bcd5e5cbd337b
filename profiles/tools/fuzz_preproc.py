"""Random sweep of vertical_band_pass / constant_space through the product path against the NumPy oracle (GPU box).
    python profiles/tools/fuzz_preproc.py <seed> <cases>"""
import sys, io, contextlib
import numpy as np
sys.path.insert(0, '.')
from impdar_amd.lib.NoInitRadarData import NoInitRadarDataFiltering
from oracle import preproc_oracle as po
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)


def mk(data, dt, dist):
    d = NoInitRadarDataFiltering()
    d.data, (d.snum, d.tnum) = data.copy(), data.shape
    d.dt, d.dist = dt, dist.copy()
    for a in ['lat', 'long', 'x_coord', 'y_coord', 'decday', 'pressure', 'elev']:
        setattr(d, a, np.cumsum(rng.random(d.tnum)))
    d.trig = np.zeros(d.tnum)
    d.trace_num = np.arange(d.tnum) + 1.
    return d


bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    ft = str(rng.choice(['butter', 'cheb', 'bessel', 'fir']))
    order = int(rng.integers(1, 7)) if ft != 'fir' else int(rng.integers(1, 200))
    padlen = 3 * (2 * order + 1) if ft != 'fir' else 0
    snum = int(rng.integers(max(padlen + 1, 2), padlen + 900))
    tnum = int(rng.integers(2, 700))
    dtype = rng.choice([np.float32, np.float64, np.int16])
    dt = float(10 ** rng.uniform(-9, -7.6))
    nyq = 0.5 / dt / 1e6
    low = float(rng.uniform(0.02, 0.3) * nyq)
    high = float(rng.uniform(0.4, 0.9) * nyq)
    raw = rng.standard_normal((snum, tnum))
    data = (raw * 3000).astype(dtype) if dtype == np.int16 else raw.astype(dtype)
    steps = 0.3 + rng.random(tnum - 1) * 1.5
    if rng.integers(0, 2):
        k = int(rng.integers(0, max(tnum - 3, 1)))
        steps[k:k + 3] = 1e-4
    dist = np.hstack(([0.], np.cumsum(steps))) / 1000. + float(rng.uniform(0, 3))
    spacing = float(rng.uniform(0.2, 5.0))
    msg = '%d %s order %d %dx%d %s dt %.1e' % (it, ft, order, snum, tnum, np.dtype(dtype).name, dt)
    try:
        d = mk(data, dt, dist)
        with contextlib.redirect_stdout(io.StringIO()):
            d.vertical_band_pass(low, high, order=order, filttype=ft)
        want = po.vertical_band_pass(data, dt, low, high, order=order, filttype=ft)
        scale = max(float(np.max(np.abs(want.astype(float)))), 1e-300)
        e1 = float(np.max(np.abs(d.data.astype(float) - want.astype(float)))) / scale
        tol1 = {np.float64: 1e-12, np.float32: 2e-7}.get(dtype, None)
        ok1 = d.data.dtype == want.dtype and (np.max(np.abs(d.data.astype(np.int64) - want.astype(np.int64))) <= 1
                                              if tol1 is None else (e1 <= tol1 or not np.isfinite(scale)))
        d = mk(data, dt, dist)
        d.constant_space(spacing)
        want2, nd, _, _ = po.constant_space(data, dist, spacing)
        e2 = float(np.max(np.abs(d.data - want2))) / max(float(np.max(np.abs(want2))), 1e-300) if want2.size else 0.0
        ok2 = d.data.shape == want2.shape and e2 < 1e-12 and np.array_equal(d.dist, nd)
    except Exception as exc:      # noqa
        ok1 = ok2 = False
        e1 = e2 = float('nan')
        msg += ' EXC %r' % (exc,)
    if not (ok1 and ok2):
        bad += 1
        print('BAD', msg, 'vbp %.2e' % e1, 'cspace %.2e' % e2)
    elif it % 10 == 9:
        print(msg, 'vbp %.1e cspace %.1e' % (e1, e2))
print('bad', bad)

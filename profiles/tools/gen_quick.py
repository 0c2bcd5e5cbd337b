#!/usr/bin/env python3
"""kirch_gen_kernel (float32, non-uniform trace spacing): parity against the C oracle on random jittered profiles
(white noise: every flipped pick shows), A/B against the ring kernel on a uniform profile, and time at config-3 size.

    python profiles/tools/gen_quick.py [ncases] [big: 0|1]
"""
import contextlib
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from impdar_amd import _hip, synth                             # noqa: E402
from impdar_amd.kirchhoff import KirchhoffPlan, migrate_resident   # noqa: E402
from impdar_amd.lib.RadarData import RadarData                  # noqa: E402
from impdar_amd.lib import migrationlib                         # noqa: E402
from oracle import c_oracle                                     # noqa: E402


def migrate(data, geo, vel, near, mode=None):
    d = RadarData(None)
    d.data, (d.snum, d.tnum) = data.copy(), data.shape
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    with contextlib.redirect_stdout(io.StringIO()):
        migrationlib.migrationKirchhoff(d, vel=vel, nearfield=near, mode=mode)
    return d.data


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    big = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(7)
    ctx = _hip.context()
    bad = 0
    worst = 0.0
    for case in range(ncases):
        snum, tnum = int(rng.integers(4, 1400)), int(rng.integers(2, 600))
        dt, dx = float(rng.choice([1e-8, 2e-9, 5e-9, 1.25e-8])), float(rng.choice([0.25, 0.5, 1.0, 2.0, 2.5]))
        vel = float(rng.choice([1.68e8, 1.69e8, 2.0e8, 1.5e8, 3.0e8]))
        t0 = float(rng.choice([0.0, dt * 1e6, -3 * dt * 1e6, 17.3 * dt * 1e6]))
        near = bool(rng.integers(0, 2))
        geo = synth.geometry(snum, tnum, dt=dt, dx=dx, t0_us=t0)
        kind = int(rng.integers(0, 4))
        if kind == 0:        # jitter around the grid
            geo['dist'] = (np.arange(tnum) + rng.uniform(-0.3, 0.3, tnum)) * dx / 1e3
        elif kind == 1:      # random steps, stationary stretches included
            steps = rng.uniform(0.3, 1.7, tnum - 1) * dx
            steps[rng.integers(0, 5, tnum - 1) == 0] = 0.0
            geo['dist'] = np.cumsum(np.concatenate([[0.], steps])) / 1e3
        elif kind == 2:      # a profile far along the line
            geo['dist'] = (50000.0 + np.cumsum(np.concatenate([[0.], rng.uniform(0.5, 1.5, tnum - 1) * dx]))) / 1e3
        # kind 3: uniform, forced through the general kernel below
        x = rng.standard_normal((snum, tnum)).astype(np.float32)
        want = c_oracle.kirchhoff(x, geo['travel_time'], geo['dist'], vel, near)
        if kind == 3:
            os.environ['IMPDAR_KIRCH_IMPL'] = 'gen'
        try:
            plan = KirchhoffPlan(ctx, np.float32, snum, tnum, geo['dist'], geo['travel_time'], vel, near, 'auto')
            kern = plan.kernel
            del plan
            got = migrate(x, geo, vel, near)
        finally:
            os.environ.pop('IMPDAR_KIRCH_IMPL', None)
        err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300)
        ok = bool(np.isfinite(got).all() and err < 1e-4)
        bad += 0 if ok else 1
        worst = max(worst, err)
        print('%3d kind %d snum %4d tnum %3d dt %.3g dx %.3g vel %.4g t0 %.3g near %d %s err %.3g %s'
              % (case, kind, snum, tnum, dt, dx, vel, t0, near, kern, err, 'ok' if ok else 'MISS'), flush=True)
    print('cases %d, misses %d, worst rel-L2 %.3g (bar 1e-4)' % (ncases, bad, worst), flush=True)

    if big:
        snum, tnum, vel = 4096, 10000, 1.69e8
        geo = synth.geometry(snum, tnum)
        jit = np.random.default_rng(3).uniform(-0.3, 0.3, tnum)
        geo_j = dict(geo)
        geo_j['dist'] = (np.arange(tnum) + jit) * 1.0 / 1e3
        x = np.random.default_rng(5).standard_normal((snum, tnum)).astype(np.float32)
        cols = np.array([0, 17, 1234, 4999, 5000, 7777, 9000, 9999])
        for name, g, env in (('jittered', geo_j, None), ('uniform via gen', geo, 'gen'), ('uniform ring', geo, None)):
            if env:
                os.environ['IMPDAR_KIRCH_IMPL'] = env
            try:
                d_in = _hip.DeviceArray.from_host(ctx, x) if hasattr(_hip, 'DeviceArray') else None
                ts = []
                out = None
                for rep in range(4):
                    t0 = time.time()
                    out, mode, ms = migrate_resident(ctx, x, g['dist'], g['travel_time'], vel, False, 'auto')
                    ts.append((time.time() - t0) * 1e3)
                    kms = ms
                print('%-16s migrate_resident: kernel ms %s, wall %s' % (name, kms, ['%.1f' % t for t in ts]), flush=True)
            finally:
                os.environ.pop('IMPDAR_KIRCH_IMPL', None)
            t0 = time.time()
            want = c_oracle.kirchhoff(x, g['travel_time'], g['dist'], vel, False, traces=cols)
            got = np.asarray(out)[:, cols]
            err = np.linalg.norm(got - want) / np.linalg.norm(want)
            print('%-16s %d spot columns vs the C oracle (%.0f s): rel-L2 %.3g, max %.3g' %
                  (name, len(cols), time.time() - t0, err, np.max(np.abs(got - want)) / np.max(np.abs(want))), flush=True)
            bad += 0 if err < 1e-4 else 1
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())

#!/usr/bin/env python3
"""One-off randomized parity sweep of the Kirchhoff paths against the C oracle (GPU box; not part of the pytest
suites).  Random sizes, spacings (round numbers included: they produce rounding-noise ties), velocities, first-sample
times, near field, float32 / float64 / int16 data, through the product entry point in its default mode.

    python tests/tools/fuzz_kirchhoff.py [ncases] [seed] [size scale]  ->  one line per case, summary at the end, exit code 1 on a miss
"""
import contextlib
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from impdar_amd import synth                                    # noqa: E402
from impdar_amd.lib.RadarData import RadarData                  # noqa: E402
from impdar_amd.lib import migrationlib                         # noqa: E402
from oracle import c_oracle                                     # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    scale_up = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    worst = {'f32': 0.0, 'f64': 0.0}
    bad = 0
    t_start = time.time()
    for case in range(ncases):
        snum, tnum = int(rng.integers(2, int(1500 * scale_up))), int(rng.integers(1, int(700 * scale_up)))
        if rng.integers(0, 2):
            dt, dx = float(rng.choice([1e-8, 2e-9, 5e-9, 1.25e-8])), float(rng.choice([0.25, 0.5, 1.0, 2.0, 2.5, 4.0]))
            vel = float(rng.choice([1.68e8, 1.69e8, 2.0e8, 1.5e8, 3.0e8]))
        else:
            dt, dx = float(10 ** rng.uniform(-9.3, -7.5)), float(10 ** rng.uniform(-1.3, 0.9))
            vel = float(rng.uniform(0.6e8, 3e8))
        t0 = float(rng.choice([0.0, dt * 1e6, -3 * dt * 1e6, 17.3 * dt * 1e6]))
        near = bool(rng.integers(0, 2))
        kind = str(rng.choice(['f32', 'f32', 'f64', 'i16']))
        geo = synth.geometry(snum, tnum, dt=dt, dx=dx, t0_us=t0)
        irregular = tnum > 2 and rng.integers(0, 5) == 0
        lattice = (not irregular) and tnum > 8 and rng.integers(0, 4) == 0
        if lattice:        # an evenly spaced survey with dropped traces: positions on the lattice, gaps of 1-3 cells; round
            # numbers make whole families of picks land ON a half-way point, which the reference decides by rounding noise
            keep = np.sort(rng.choice(int(tnum * 1.25) + 2, size=tnum, replace=False))
            geo['dist'] = (keep - keep[0]) * dx / 1e3
            if rng.integers(0, 3) == 0:
                geo['dist'] = geo['dist'] + float(rng.choice([2.0, 50.0, 1234.5]))      # ... starting far along the line
        if irregular:      # uneven trace spacing (and, half of the time, an uneven time axis): the per-pair kernels
            geo['dist'] = np.cumsum(np.concatenate([[0.], rng.uniform(0.3, 1.7, tnum - 1) * dx])) / 1e3
            if rng.integers(0, 2) and snum > 2:
                tt = geo['travel_time'].copy()
                tt[1:] += np.cumsum(rng.uniform(-0.2, 0.2, snum - 1)) * dt * 1e6 * 0.1
                geo['travel_time'] = np.sort(tt)
        x = rng.standard_normal((snum, tnum))
        data = {'f32': x.astype(np.float32), 'f64': x, 'i16': (x * 1000).astype(np.int16)}[kind]
        want = c_oracle.kirchhoff(data, geo['travel_time'], geo['dist'], vel, near)
        d = RadarData(None)
        d.data, d.snum, d.tnum = data.copy(), snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        with contextlib.redirect_stdout(io.StringIO()):
            migrationlib.migrationKirchhoff(d, vel=vel, nearfield=near)
        scale = max(np.max(np.abs(want)), 1e-300)
        if kind == 'f32':
            err = np.linalg.norm(d.data - want) / max(np.linalg.norm(want), 1e-300)
            tol, key = 1e-4, 'f32'
        else:
            err = np.max(np.abs(d.data - want)) / scale
            tol, key = 1e-12, 'f64'
        ok = bool(np.isfinite(d.data).all() == np.isfinite(want).all() and err < tol)
        bad += 0 if ok else 1
        worst[key] = max(worst[key], err)
        print('%3d %s snum %4d tnum %3d dt %.3g dx %.3g vel %.4g t0 %.3g near %d %s err %.3g %s'
              % (case, kind, snum, tnum, dt, dx, vel, t0, near, 'irregular' if irregular else ('lattice' if lattice else 'uniform'), err,
                 'ok' if ok else 'MISS'), flush=True)
    print('cases %d, misses %d, worst float32 rel-L2 %.3g (bar 1e-4), worst float64/int16 rel-max %.3g (bar 1e-12), %.0f s'
          % (ncases, bad, worst['f32'], worst['f64'], time.time() - t_start))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())

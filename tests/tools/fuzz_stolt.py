#!/usr/bin/env python3
"""One-off randomized parity sweep of the Stolt f-k path against the NumPy oracle (GPU box; not part of the pytest
suites).  Random sizes (odd and even, primes included: rocFFT takes any length), spacings, velocities, tapers,
float32 / float64 / int16 data.

    python tests/tools/fuzz_stolt.py [ncases] [seed]  ->  one line per case, summary at the end, exit code 1 on a miss
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
from oracle import mig_oracle                                   # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = {'f32': 0.0, 'f64': 0.0}
    bad = 0
    t_start = time.time()
    for case in range(ncases):
        snum, tnum = int(rng.integers(2, 1400)), int(rng.integers(1, 900))
        dt, dx = float(rng.choice([1e-8, 2e-9, 5e-9, 1.25e-8])), float(rng.choice([0.3, 0.5, 1.0, 2.0, 2.5, 4.0]))
        vel = float(rng.choice([1.68e8, 1.69e8, 2.0e8, 1.5e8, 3.0e8]))
        ht, vt = int(rng.integers(1, 120)), int(rng.integers(1, 1200))
        kind = str(rng.choice(['f32', 'f32', 'f64', 'i16']))
        geo = synth.geometry(snum, tnum, dt=dt, dx=dx)
        x = rng.standard_normal((snum, tnum))
        data = {'f32': x.astype(np.float32), 'f64': x, 'i16': (x * 1000).astype(np.int16)}[kind]
        want = mig_oracle.stolt(data, geo['dt'], geo['trace_int'], geo['dist'], vel, ht, vt)
        d = RadarData(None)
        d.data, d.snum, d.tnum = data.copy(), snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        with contextlib.redirect_stdout(io.StringIO()):
            migrationlib.migrationStolt(d, vel=vel, htaper=ht, vtaper=vt)
        got = np.asarray(d.data, dtype=np.float64)
        want = np.asarray(want, dtype=np.float64)
        if got.shape != want.shape:
            err, ok, key = float('inf'), False, 'f64'
        elif kind == 'f32':
            err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300)
            ok, key = bool(np.isfinite(got).all() and err < 1e-4), 'f32'
        else:
            err = np.max(np.abs(got - want)) / max(np.max(np.abs(want)), 1e-300) if got.size else 0.0
            ok, key = bool(np.isfinite(got).all() and err < 1e-12), 'f64'
        bad += 0 if ok else 1
        worst[key] = max(worst[key], err if np.isfinite(err) else 1e300)
        print('%3d %s snum %4d tnum %3d dt %.3g dx %.3g vel %.4g tapers %d %d err %.3g %s'
              % (case, kind, snum, tnum, dt, dx, vel, ht, vt, err, 'ok' if ok else 'MISS'), flush=True)
    print('cases %d, misses %d, worst float32 rel-L2 %.3g (bar 1e-4), worst float64/int16 rel-max %.3g (bar 1e-12), %.0f s'
          % (ncases, bad, worst['f32'], worst['f64'], time.time() - t_start))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())

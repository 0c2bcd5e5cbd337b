#!/usr/bin/env python3
"""One-off sweep of the pipelined one-shot Kirchhoff call (GPU box; not part of the pytest suites): random large
geometries -- trace counts off the 8-trace groups, apertures from a tenth of the profile to all of it, float32 and
float64, far and near field -- migrated in pieces (the default: upload in trace chunks, several launches, downloads
underneath), with IMPDAR_KIRCH_ONESHOT_SPLIT=2 (one upload, two launches) and =0 (one upload, one launch, one download):
the three must be BIT-EQUAL.

    python tests/tools/fuzz_oneshot_pieces.py [ncases] [seed]
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


def run(x, geo, snum, tnum, vel, nearfield, split):
    if split:
        os.environ['IMPDAR_KIRCH_ONESHOT_SPLIT'] = split
    else:
        os.environ.pop('IMPDAR_KIRCH_ONESHOT_SPLIT', None)
    d = RadarData(None)
    d.data, d.snum, d.tnum = x, snum, tnum
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    with contextlib.redirect_stdout(io.StringIO()):
        d.migrate('kirch', vel=vel, nearfield=nearfield)
    return d.data


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    t_start = time.time()
    for case in range(ncases):
        dtype = np.float32 if rng.random() < 0.6 else np.float64
        snum = int(rng.choice([2048, 3000, 4096]))
        lo = 4096 if dtype == np.float64 or snum == 4096 else 8192          # the pieces start at 64 MB
        tnum = int(rng.integers(lo, lo + 6000))
        dx = float(rng.choice([1.0, 2.0, 4.0, 8.0, 16.0]))
        dt = float(rng.choice([1e-8, 5e-9]))
        vel = float(rng.choice([1.69e8, 1.68e8, 2.0e8]))
        nearfield = bool(rng.random() < 0.25)
        geo = synth.geometry(snum, tnum, dt=dt, dx=dx)
        x = synth.noise_radargram(snum, tnum, seed=int(rng.integers(1 << 30))).astype(dtype)
        a = run(x, geo, snum, tnum, vel, nearfield, '')
        c = run(x, geo, snum, tnum, vel, nearfield, '2')
        b = run(x, geo, snum, tnum, vel, nearfield, '0')
        ok = np.array_equal(a, b) and np.array_equal(c, b) and np.isfinite(a).all() and a.any()
        bad += 0 if ok else 1
        print('%3d %s snum %4d tnum %5d dx %4.1f dt %.0e vel %.3g near %d %s'
              % (case, np.dtype(dtype).name, snum, tnum, dx, dt, vel, nearfield, 'bit-equal' if ok else 'MISS'), flush=True)
    print('cases %d, misses %d, %.0f s' % (ncases, bad, time.time() - t_start))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()

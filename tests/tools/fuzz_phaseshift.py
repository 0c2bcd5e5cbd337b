#!/usr/bin/env python3
"""One-off randomized parity sweep of the phase-shift paths against the NumPy oracle (GPU box; not part of the
pytest suites: the oracle needs seconds per case).  Random sizes, sample / trace spacings, layered tables (thin and
thick layers, first layers at round velocities so that boundary frequencies occur), float32 and float64 data.

    python tests/tools/fuzz_phaseshift.py [ncases] [seed]  ->  one line per case, summary at the end, exit code 1 on a miss

IMPDAR_PS_MFMA=7 in the environment sends every v(z) case to ps_series_kernel, =6 every table to ps_nufft_kernel (round 6); the
kernel that ran is printed with each case.
"""
import contextlib
import ctypes
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from impdar_amd import synth, _hip                              # noqa: E402
from impdar_amd.lib.RadarData import RadarData                  # noqa: E402
from impdar_amd.lib import migrationlib                         # noqa: E402
from oracle import mig_oracle                                   # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = {np.float32: 0.0, np.float64: 0.0}
    bad = 0
    kernels = {}
    t_start = time.time()
    for case in range(ncases):
        snum = int(rng.choice([40, 97, 128, 200, 333, 512, 700, 1000, 1300, 2100]))
        tnum = int(rng.choice([16, 33, 64, 100, 256, 512, 600])) if snum < 1000 else int(rng.choice([16, 33, 64]))
        dt = float(rng.choice([1e-8, 2e-9, 5e-9, 1.25e-8]))
        dx = float(rng.choice([1.0, 0.5, 2.0, 2.5, 4.0]))
        geo = synth.geometry(snum, tnum, dt=dt, dx=dx)
        Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
        kind = rng.integers(0, 5)
        if kind == 0:
            vel = float(rng.choice([1.68e8, 1.69e8, 2.0e8, 1.5e8]))
        elif kind == 4:
            # "per-step": a dense table (a row per few samples, gradient + wobble + small noise, rising or falling), so
            # that the interpolated velocity changes at EVERY depth step (ps_smooth_kernel / ps_smooth32_kernel)
            nl = int(min(max(snum // int(rng.choice([1, 2, 3])), 8), 700))
            v0 = float(rng.choice([1.68e8, 1.69e8, 2.0e8, 1.6e8]))
            u = np.linspace(0., 1., nl)
            vs = v0 * (1.0 + float(rng.uniform(-0.15, 0.3)) * u + 0.03 * np.sin(float(rng.uniform(2., 12.)) * u)
                       + float(rng.choice([0., 1e-4])) * rng.standard_normal(nl))
            Rv = vs.max() * geo['travel_time'][-1] * 1e-6 / 2.
            zs = np.linspace(0., 1.3 * Rv, nl)
            vel = np.stack([vs, zs], axis=1)
        else:
            nl = int(rng.choice([2, 3, 4, 7, 40])) if kind < 3 else int(rng.integers(50, 90))
            v0 = float(rng.choice([1.68e8, 1.69e8, 2.0e8, 1.6e8]))
            vs = np.concatenate([[v0], v0 + np.cumsum(rng.uniform(0.0, 0.06e8, nl - 1))])
            Rv = vs.max() * geo['travel_time'][-1] * 1e-6 / 2.
            zs = np.sort(np.concatenate([[0.], rng.uniform(0.05, 1.2, nl - 1)])) * Rv
            zs[-1] = 1.3 * Rv
            vel = np.stack([vs, zs], axis=1)
        dtype = np.float32 if rng.integers(0, 3) else np.float64
        data = synth.noise_radargram(snum, tnum, seed=int(rng.integers(1 << 30))).astype(dtype)
        ht, vt = int(rng.integers(1, 30)), int(rng.integers(1, 30))
        try:
            want = mig_oracle.phase_shift(data.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'],
                                          geo['dist'], vel, ht, vt)
        except ValueError as exc:
            # a table the reference refuses (it does not cover the profile's depths): the product must refuse it too
            d = RadarData(None)
            d.data, d.snum, d.tnum = data.copy(), snum, tnum
            d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    migrationlib.migrationPhaseShift(d, vel=vel, htaper=ht, vtaper=vt)
                refused = False
            except ValueError:
                refused = True
            bad += 0 if refused else 1
            print('%3d table refused by the oracle (%s): product %s' % (case, exc, 'refuses too' if refused else 'MISS'), flush=True)
            continue
        d = RadarData(None)
        d.data, d.snum, d.tnum = data.copy(), snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        with contextlib.redirect_stdout(io.StringIO()):
            migrationlib.migrationPhaseShift(d, vel=vel, htaper=ht, vtaper=vt)
        buf = ctypes.create_string_buffer(1024)
        _hip.load().impdar_ctx_last_metrics(_hip.context(), buf, len(buf))
        kernel = json.loads(buf.value.decode()).get('kernel', '?')
        kernels[kernel] = kernels.get(kernel, 0) + 1
        if dtype == np.float32:
            err = np.linalg.norm(d.data - want) / max(np.linalg.norm(want), 1e-300)
            tol = 2e-4
        else:
            err = np.max(np.abs(d.data - want)) / max(np.max(np.abs(want)), 1e-300)
            tol = 1e-10
        ok = np.isfinite(d.data).all() and err < tol
        bad += 0 if ok else 1
        worst[dtype] = max(worst[dtype], err)
        print('%3d %s snum %4d tnum %3d dt %.3g dx %.3g vel %s %s err %.3g %s'
              % (case, 'f32' if dtype == np.float32 else 'f64', snum, tnum, dt, dx,
                 'const %.3g' % vel if np.isscalar(vel) else '%s%d layers from %.3g' % ('per-step ' if kind == 4 else '', len(vel), vel[0, 0]),
                 kernel, err, 'ok' if ok else 'MISS'), flush=True)
    print('cases %d, misses %d, worst float32 rel-L2 %.3g (bar 2e-4), worst float64 rel-max %.3g (bar 1e-10), %.0f s'
          % (ncases, bad, worst[np.float32], worst[np.float64], time.time() - t_start))
    print('kernels: ' + ', '.join('%s %d' % kv for kv in sorted(kernels.items())))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())

#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench contract): Stolt f-k at
BASELINE config 2 (4096x4096 float32), phase-shift at config 5 (8192x8192,
constant v and 1-D v(z)), the v(x,z) finite-difference branch (256x512) and the band-pass / re-spacing steps in front of a
migration at config-3 size (4096x10000 float32, resident in HBM, plus the
three-step chain with and without residency), each through the product path on
one MI355X.
Prints one JSON line per path.  Host wall time includes H2D/D2H of the
radargram (the entry points take host buffers).  Each line carries a
``cpu_baseline``: the NumPy oracle (a port of the reference's algorithm in
closed form, far faster than the reference's Python loops) timed on the host
cores of the same box on a bounded sample, with the sample stated.  Kernel
times: run under ``rocprofv3 --kernel-trace --stats`` (profiles/r01_paths_*)."""
import argparse
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--stolt', type=int, default=4096)
    ap.add_argument('--phsh', type=int, default=8192)
    ap.add_argument('--chain', type=str, default='4096x10000', help='snum x tnum of the band-pass / re-spacing lines')
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--skip', default='')
    ap.add_argument('--no-cpu', action='store_true')
    args = ap.parse_args()
    from impdar_amd import _hip, synth
    from impdar_amd.lib.RadarData import RadarData
    from impdar_amd.lib import migrationlib
    import contextlib
    import io
    _hip.load()
    assert _hip.device_count() > 0

    def dat_of(data, geo):
        d = RadarData(None)
        d.data, (d.snum, d.tnum) = data, data.shape
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        return d

    def timed(fn, make):
        best = None
        for _ in range(args.reps + 1):           # first call pays rocFFT plan creation
            d = make()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                fn(d)
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        return best, d

    def cpu(fn, label):
        if args.no_cpu:
            return None
        from oracle import mig_oracle                      # the checker, timed as the CPU baseline
        t0 = time.perf_counter()
        fn(mig_oracle)
        el = time.perf_counter() - t0
        return {"seconds": el, "kind": "port", "cores": 1, "sample": label}

    rng = np.random.default_rng(0)
    if 'stolt' not in args.skip:
        n = args.stolt
        geo = synth.geometry(n, n)
        x = rng.standard_normal((n, n)).astype(np.float32)
        el, d = timed(lambda d: migrationlib.migrationStolt(d, htaper=100, vtaper=1000), lambda: dat_of(x.copy(), geo))
        cb = cpu(lambda o: o.stolt(x, geo['dt'], geo['trace_int'], geo['dist'], 1.68e8, 100, 1000),
                 "NumPy oracle, same %dx%d float32 radargram, one process" % (n, n))
        print(json.dumps({"path": "stolt", "config": "%dx%d float32 (BASELINE config 2)" % (n, n),
                          "host_seconds": el, "traces_per_s": n / el, "finite": bool(np.isfinite(d.data).all()),
                          "algorithmic_bytes": 40 * n * n, "reference_seconds_same_size": 100.30,
                          "cpu_baseline": cb}), flush=True)
    if 'phsh' not in args.skip:
        n = args.phsh
        geo = synth.geometry(n, n)
        x = rng.standard_normal((n, n)).astype(np.float32)
        el, d = timed(lambda d: migrationlib.migrationPhaseShift(d, vel=1.69e8), lambda: dat_of(x.copy(), geo))
        m = min(n, 1024)
        gs = synth.geometry(m, m)
        xs = x[:m, :m].astype(np.float64)
        cb = cpu(lambda o: o.phase_shift(xs, gs['dt'], gs['trace_int'], gs['travel_time'], gs['dist'], 1.69e8),
                 "NumPy oracle on a %dx%d corner (work scales with snum*nt*tnum: x%d for the full size)"
                 % (m, m, (n // m) ** 3))
        print(json.dumps({"path": "phase-shift const v", "config": "%dx%d float32" % (n, n), "host_seconds": el,
                          "traces_per_s": n / el, "finite": bool(np.isfinite(d.data).all()),
                          "rotate_accumulate_steps": float(n) ** 3, "cpu_baseline": cb}), flush=True)
        Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
        tab = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
        el, d = timed(lambda d: migrationlib.migrationPhaseShift(d, vel=tab), lambda: dat_of(x.copy(), geo))
        m = min(n, 512)
        gs = synth.geometry(m, m)
        xs = x[:m, :m].astype(np.float64)
        Rs = 1.9e8 * gs['travel_time'][-1] * 1e-6 / 2.
        tabs = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rs], [1.8e8, 0.5 * Rs], [1.9e8, 1.2 * Rs]])
        cb = cpu(lambda o: o.phase_shift(xs, gs['dt'], gs['trace_int'], gs['travel_time'], gs['dist'], tabs),
                 "NumPy oracle on a %dx%d radargram (work scales with snum*nt*tnum: x%d for the full size)"
                 % (m, m, (n // m) ** 3))
        print(json.dumps({"path": "phase-shift v(z) Gazdag", "config": "%dx%d float32 (BASELINE config 5)" % (n, n),
                          "host_seconds": el, "traces_per_s": n / el, "finite": bool(np.isfinite(d.data).all()),
                          "rotate_accumulate_steps": float(n) ** 3, "reference_extrapolated_hours": 7.6,
                          "cpu_baseline": cb}), flush=True)
    if 'phsh64' not in args.skip:
        # the same layered table on float64 data (what a float64 .mat file gets): ps_vz64_kernel
        n = args.phsh
        geo = synth.geometry(n, n)
        x64 = rng.standard_normal((n, n))
        Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
        tab = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
        el, d = timed(lambda d: migrationlib.migrationPhaseShift(d, vel=tab), lambda: dat_of(x64.copy(), geo))
        print(json.dumps({"path": "phase-shift v(z) Gazdag, float64 data", "config": "%dx%d float64" % (n, n),
                          "host_seconds": el, "traces_per_s": n / el, "finite": bool(np.isfinite(d.data).all()),
                          "rotate_accumulate_steps": float(n) ** 3}), flush=True)
        del x64, d
    if 'ffd' not in args.skip:
        # v(x,z) table: the Fourier finite-difference branch, a serial chain of snum * nt steps (one workgroup, the
        # row in LDS, for power-of-two trace counts up to 512)
        sn, tn = 256, 512
        geo = synth.geometry(sn, tn, dx=5.0)
        x = rng.standard_normal((sn, tn))
        Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
        xs_ = np.linspace(0., geo['dist'][-1] * 1e3, 5)
        tab3 = np.array([(v + (2e5 * xx / xs_[-1] if z else 0.), z * Rp, xx) for xx in xs_
                         for v, z in ((1.69e8, 0.), (1.72e8, 0.6), (1.8e8, 1.3))])
        el, d = timed(lambda d: migrationlib.migrationPhaseShift(d, vel=tab3, htaper=10, vtaper=10), lambda: dat_of(x.copy(), geo))
        nt_ = 1 << int(np.ceil(np.log2(sn)))
        ms_, mt_ = 32, 64
        gs = synth.geometry(ms_, mt_, dx=5.0)
        Rs = 1.9e8 * gs['travel_time'][-1] * 1e-6 / 2.
        xq = np.linspace(0., gs['dist'][-1] * 1e3, 5)
        tabs3 = np.array([(v + (2e5 * xx / xq[-1] if z else 0.), z * Rs, xx) for xx in xq
                          for v, z in ((1.69e8, 0.), (1.72e8, 0.6), (1.8e8, 1.3))])
        cb = cpu(lambda o: o.phase_shift(x[:ms_, :mt_], gs['dt'], gs['trace_int'], gs['travel_time'], gs['dist'], tabs3, 10, 10),
                 "NumPy oracle on a %dx%d radargram: %d chain steps of %d traces (the full size has %d steps of %d)"
                 % (ms_, mt_, ms_ * ms_, mt_, sn * nt_, tn))
        print(json.dumps({"path": "phase-shift v(x,z) Fourier finite-difference", "config": "%dx%d float64" % (sn, tn),
                          "host_seconds": el, "chain_steps": sn * nt_, "us_per_step": el / (sn * nt_) * 1e6,
                          "finite": bool(np.isfinite(d.data).all()), "cpu_baseline": cb}), flush=True)
    if 'chain' not in args.skip:
        from impdar_amd import preproc
        from impdar_amd.lib.NoInitRadarData import NoInitRadarDataFiltering
        snum, tnum = (int(v) for v in args.chain.split('x'))
        ctx, lib = _hip.context(), _hip.load()
        x = rng.standard_normal((snum, tnum)).astype(np.float32)
        dist = np.hstack(([0.], np.cumsum(0.6 + 0.8 * rng.random(tnum - 1)))) / 1000.

        def dev_ms(fn, reps=10):
            fn()
            lib.impdar_ctx_sync(ctx)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            lib.impdar_ctx_sync(ctx)
            return (time.perf_counter() - t0) / reps * 1e3

        def sample_cpu(fn, label):
            if args.no_cpu:
                return None
            t0 = time.perf_counter()
            fn()
            return {"seconds": time.perf_counter() - t0, "kind": "reference", "cores": 1, "sample": label}

        d_x = _hip.DeviceArray.from_host(ctx, x)
        spec = preproc.design_filter(1e-8, 2., 10.)
        ms = dev_ms(lambda: preproc.filter_dev(d_x, spec))
        L = snum + 6 * len(spec[1])
        algo = 2 * snum * tnum * 4 + 2 * L * tnum * 8        # read x, write/read the fp64 forward pass, write out
        m = min(tnum, 1000)
        from scipy import signal
        cb = sample_cpu(lambda: signal.filtfilt(spec[1], spec[2], x[:, :m], axis=0).astype(np.float32),
                        "scipy.signal.filtfilt (what the reference calls) on %d of %d traces" % (m, tnum))
        print(json.dumps({"path": "vertical_band_pass butter order 5 (filtfilt), resident", "config": "%dx%d float32" % (snum, tnum),
                          "device_ms": ms, "traces_per_s": tnum / ms * 1e3, "algorithmic_bytes": algo,
                          "roofline": {"bound": "hbm", "achieved": algo / ms / 1e6, "peak": 8000.0, "unit": "GB/s",
                                       "frac": algo / ms / 1e6 / 8000.0,
                                       "note": "serial fp64 recurrence along time, 21 non-fused operations per sample: "
                                               "issue-bound at 625 wavefronts, not HBM-bound"},
                          "cpu_baseline": cb}), flush=True)
        d_x.free()
        d_x = _hip.DeviceArray.from_host(ctx, x)
        plan = preproc.SpacingPlan(dist.copy(), 1.0)

        def lerp():
            o = plan.apply_dev(d_x)
            lib.impdar_ctx_sync(ctx)
            o.free()
        ms = dev_ms(lerp)
        algo = snum * tnum * 4 + snum * plan.n_new * 8
        from scipy.interpolate import interp1d
        ms_rows = min(snum, 256)
        cb = sample_cpu(lambda: interp1d(dist, x[:ms_rows])(plan.new_dists),
                        "scipy.interpolate.interp1d (what the reference calls) on %d of %d sample rows" % (ms_rows, snum))
        print(json.dumps({"path": "constant_space (linear re-spacing), resident", "config": "%dx%d float32 -> %d traces float64" % (snum, tnum, plan.n_new),
                          "device_ms": ms, "traces_per_s": tnum / ms * 1e3, "algorithmic_bytes": algo,
                          "roofline": {"bound": "hbm", "achieved": algo / ms / 1e6, "peak": 8000.0, "unit": "GB/s",
                                       "frac": algo / ms / 1e6 / 8000.0,
                                       "note": "includes the output allocation and the upload of the gather tables"},
                          "cpu_baseline": cb}), flush=True)
        d_x.free()

        def chain(resident):
            d = NoInitRadarDataFiltering()
            d.data, (d.snum, d.tnum) = x.copy(), x.shape
            d.dt, d.dist = 1e-8, dist.copy()
            d.travel_time = np.arange(snum) * 1e-2
            for a in ['lat', 'long', 'x_coord', 'y_coord', 'decday', 'pressure', 'elev']:
                setattr(d, a, np.arange(tnum, dtype=float))
            d.trig = np.zeros(tnum)
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                if resident:
                    d.to_device()
                d.vertical_band_pass(2., 10.)
                d.constant_space(1.0)
                d.migrate('stolt', htaper=100, vtaper=1000)
                if resident:
                    d.from_device()
            return time.perf_counter() - t0, d
        for resident in (False, True):
            best = min(chain(resident)[0] for _ in range(args.reps + 1))
            print(json.dumps({"path": "chain vbp -> constant_space -> stolt, %s" % ("resident in HBM" if resident else "host buffers between steps"),
                              "config": "%dx%d float32 in, float64 out" % (snum, tnum), "host_seconds": best,
                              "traces_per_s": tnum / best}), flush=True)


if __name__ == '__main__':
    main()

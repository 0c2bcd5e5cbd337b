"""Where the runs kernels stop paying: device ms at n x n for layered tables with a growing share of tiles that hold
a velocity change, runs kernel forced (IMPDAR_PS_DIRTY_MAX=1) against per-step kernel forced (=0), float32 and float64."""
import sys, os, json, io, contextlib, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import ctypes as C
    import numpy as np
    from impdar_amd import _hip, synth
    from impdar_amd.lib.RadarData import RadarData
    n, nl, dt = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    lib, ctx = _hip.load(), _hip.context()
    geo = synth.geometry(n, n)
    x = np.random.default_rng(0).standard_normal((n, n)).astype(dt)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    vel = np.stack([np.linspace(1.69e8, 1.9e8, nl), np.linspace(0., 1.2 * Rp, nl)], axis=1)
    ms = []
    for i in range(3):
        d = RadarData(None)
        d.data, (d.snum, d.tnum) = x, x.shape
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        d.to_device()
        with contextlib.redirect_stdout(io.StringIO()):
            d.migrate('phsh', vel=vel, htaper=100, vtaper=1000)
        v = C.c_float()
        _hip.check(lib.impdar_ctx_last_ms(ctx, C.byref(v)), 'last_ms')
        ms.append(v.value)
        d._dev.free(); d._dev = None
    print(min(ms[1:]))
    sys.exit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for dt in ('float32', 'float64'):
    for nl in (4, 16, 40, 80, 160):
        row = []
        for knob in ('1', '0'):
            env = dict(os.environ, IMPDAR_PS_DIRTY_MAX=knob)
            out = subprocess.run([sys.executable, __file__, 'child', str(n), str(nl), dt], env=env, capture_output=True, text=True)
            row.append(float(out.stdout.strip().splitlines()[-1]))
        print('%s n %d layers %3d (about %2.0f %% of the %d tiles hold a change): runs kernel %.2f ms, per-step kernel %.2f ms'
              % (dt, n, nl, 100. * min(1., 5. * (nl - 1) / 1.2 / (n / 16)), n // 16, row[0], row[1]), flush=True)

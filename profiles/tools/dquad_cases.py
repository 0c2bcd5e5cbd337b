import os, sys
import numpy as np
sys.path.insert(0, '.')
from impdar_amd import synth, _hip
from impdar_amd.kirchhoff import KirchhoffPlan
from oracle import c_oracle

def run(impl, x, geo, vel, near, xbd=None):
    for k, v in (('IMPDAR_KIRCH_EXACT_IMPL', impl), ('IMPDAR_KIRCH_XBD', xbd)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    ctx = _hip.context()
    snum, tnum = x.shape
    plan = KirchhoffPlan(ctx, x.dtype, snum, tnum, geo['dist'], geo['travel_time'], vel, near, 'exact')
    d_in = _hip.DeviceArray.from_host(ctx, x); d_out = _hip.DeviceArray(ctx, (snum, tnum), x.dtype)
    plan.prep(d_in, tnum, 0, tnum); plan.migrate(d_out, 0, tnum); plan.sync()
    out = d_out.to_host(); plan.destroy(); d_in.free(); d_out.free()
    return out

rng = np.random.default_rng(7)
for case in range(10):
    snum, tnum = int(rng.integers(40, 900)), int(rng.integers(3, 300))
    dt, dx = float(rng.choice([2e-9, 5e-9, 1e-8])), float(rng.choice([0.3, 1.0, 2.5, 6.0]))
    vel, t0 = float(rng.choice([1.2e8, 1.69e8, 3e8])), float(rng.choice([0.0, 0.004, -0.02]))
    near = bool(rng.integers(0, 2))
    geo = synth.geometry(snum, tnum, dt=dt, dx=dx, t0_us=t0)
    x = synth.noise_radargram(snum, tnum, seed=100 + case)
    want = c_oracle.kirchhoff(x, geo['travel_time'], geo['dist'], vel, near)
    sa = 2 * dx / (vel * dt)
    errs = []
    for impl, xbd in ((None, '20'), (None, '16'), ('tab', None)):
        got = run(impl, x, geo, vel, near, xbd)
        e = np.abs(got - want) / np.max(np.abs(want))
        bad = np.argwhere(e > 1e-12)
        errs.append((float(e.max()), len(bad), bad[:3].tolist(), bad[-2:].tolist()))
    print(case, snum, tnum, dt, dx, vel, t0, near, 'sa=%.2f' % sa, errs, flush=True)

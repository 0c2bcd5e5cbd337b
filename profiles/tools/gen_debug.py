#!/usr/bin/env python3
"""Where does kirch_gen_kernel differ from the oracle?  (diagnostic)"""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from impdar_amd import _hip, synth
from impdar_amd.lib.RadarData import RadarData
from impdar_amd.lib import migrationlib
from oracle import c_oracle

def migrate(data, geo, vel, near):
    d = RadarData(None)
    d.data, (d.snum, d.tnum) = data.copy(), data.shape
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    with contextlib.redirect_stdout(io.StringIO()):
        migrationlib.migrationKirchhoff(d, vel=vel, nearfield=near)
    return d.data

rng = np.random.default_rng(11)
for (snum, tnum, dt, dx, vel, t0, near, kind) in [(79, 13, 5e-9, 1.0, 2e8, 0.0, False, 1), (79, 13, 5e-9, 1.0, 2e8, 0.0, False, 0),
                                                  (245, 4, 1.25e-8, 2.5, 1.69e8, 0.0, False, 2), (300, 40, 1e-8, 1.0, 1.69e8, 0.0, False, 1),
                                                  (300, 40, 1e-8, 1.0, 1.69e8, 0.0, False, 4)]:
    geo = synth.geometry(snum, tnum, dt=dt, dx=dx, t0_us=t0)
    if kind == 0:
        geo['dist'] = (np.arange(tnum) + rng.uniform(-0.3, 0.3, tnum)) * dx / 1e3
    elif kind == 1:
        steps = rng.uniform(0.3, 1.7, tnum - 1) * dx
        steps[rng.integers(0, 5, tnum - 1) == 0] = 0.0
        geo['dist'] = np.cumsum(np.concatenate([[0.], steps])) / 1e3
    elif kind == 2:
        geo['dist'] = (50000.0 + np.cumsum(np.concatenate([[0.], rng.uniform(0.5, 1.5, tnum - 1) * dx]))) / 1e3
    elif kind == 4:
        steps = rng.uniform(0.3, 1.7, tnum - 1) * dx
        geo['dist'] = np.cumsum(np.concatenate([[0.], steps])) / 1e3
    x = rng.standard_normal((snum, tnum)).astype(np.float32)
    want = c_oracle.kirchhoff(x, geo['travel_time'], geo['dist'], vel, near)
    got = migrate(x, geo, vel, near)
    diff = np.abs(got - want)
    print('case', snum, tnum, kind, 'rel-L2 %.3g' % (np.linalg.norm(got - want) / np.linalg.norm(want)))
    bad = np.argwhere(diff > 1e-5 * np.max(np.abs(want)))
    print('  bad entries', len(bad), 'of', diff.size)
    if len(bad):
        rows = np.bincount(bad[:, 0], minlength=snum); cols = np.bincount(bad[:, 1], minlength=tnum)
        print('  rows with errors:', np.nonzero(rows)[0][:40], '...', 'cols:', np.nonzero(cols)[0][:40])
        for (r, c) in bad[:6]:
            print('   (ti %d, xi %d): got %.6g want %.6g' % (r, c, got[r, c], want[r, c]))
        print('  dist steps (m):', np.round(np.diff(geo['dist']) * 1e3, 3)[:20])

#!/usr/bin/env python3
"""kirch_gen_kernel at config-3 size on a jittered profile: kernel time of a few launches (for rocprofv3)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from impdar_amd import _hip, synth
from impdar_amd.kirchhoff import KirchhoffPlan

snum, tnum, vel = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 10000, 1.69e8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
geo = synth.geometry(snum, tnum)
geo['dist'] = (np.arange(tnum) + np.random.default_rng(3).uniform(-0.3, 0.3, tnum)) / 1e3
x = np.random.default_rng(5).standard_normal((snum, tnum)).astype(np.float32)
ctx = _hip.context()
plan = KirchhoffPlan(ctx, np.float32, snum, tnum, geo['dist'], geo['travel_time'], vel, False, 'auto')
print('kernel', plan.kernel)
d_in = _hip.DeviceArray.from_host(ctx, x)
d_out = _hip.DeviceArray(ctx, (snum, tnum), np.float32)
for r in range(reps):
    plan.prep(d_in, tnum, 0, tnum)
    plan.migrate(d_out, 0, tnum)
    plan.sync()
    print('launch %d: prep/gather/migrate ms' % r, plan.last_ms())

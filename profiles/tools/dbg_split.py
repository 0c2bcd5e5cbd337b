import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from impdar_amd import synth, _hip, parallel
from impdar_amd.kirchhoff import KirchhoffPlan
snum, tnum, vel = 4096, int(sys.argv[1]), 1.69e8
geo = synth.geometry(snum, tnum)
ctx = _hip.context()
x = np.random.default_rng(0).standard_normal((snum, tnum)).astype(np.float32)
d_in = _hip.DeviceArray.from_host(ctx, x)
res = {}
lo, hi = int(sys.argv[2]), int(sys.argv[3])
for split in ('0', '1'):
    os.environ['IMPDAR_KIRCH_SPLIT'] = split
    os.environ['IMPDAR_KIRCH_XB'] = '40'; os.environ['IMPDAR_KIRCH_NH'] = '1'
    plan = KirchhoffPlan(ctx, np.float32, snum, tnum, geo['dist'], geo['travel_time'], vel, mode='fast', nranks=1)
    d = _hip.DeviceArray(ctx, (snum, hi - lo), np.float32)
    plan.prep(d_in, tnum, 0, tnum); plan.migrate(d, lo, hi); plan.sync()
    res[split] = d.to_host(); d.free(); plan.destroy()
a, b = res['0'].astype(np.float64), res['1'].astype(np.float64)
err = np.abs(a - b)
print('rel l2', np.linalg.norm(a - b) / np.linalg.norm(a), 'max', err.max() / np.abs(a).max())
bad = np.argwhere(err > 1e-5 * np.abs(a).max())
print('bad count', len(bad))
if len(bad):
    rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
    print('rows', rows.min(), rows.max(), len(rows), 'cols', cols.min(), cols.max(), len(cols))
    print('col histogram (per 40):', np.bincount(bad[:, 1] // 40)[:60])
    print('row histogram (per 256):', np.bincount(bad[:, 0] // 256))

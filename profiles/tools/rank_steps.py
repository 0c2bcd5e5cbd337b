#!/usr/bin/env python3
"""Emulate the per-rank steady state of an N-rank strong-scaling run on ONE GPU: for every rank's
(input shard, output block) run prep + migrate back to back (pipelined, as bench.py does) and report the
sustained step time.  The all-gather is not emulated.  Usage: rank_steps.py [N ...]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from impdar_amd import _hip, parallel, synth
from impdar_amd.kirchhoff import KirchhoffPlan
snum, tnum, vel = 4096, 10000, 1.69e8
geo = synth.geometry(snum, tnum); tt = geo['travel_time'] / 1e6
_hip.load(); ctx = _hip.context(0)
data = np.random.default_rng(0).standard_normal((snum, tnum)).astype(np.float32)
plans = {}


def plan_for(n):          # a plan built for n ranks uses the shard (LDS-free) prep kernel for n > 1
    if n not in plans:
        plans[n] = KirchhoffPlan(ctx, np.float32, snum, tnum, geo['dist'], geo['travel_time'], vel, False, 'fast', nranks=n)
    return plans[n]

def sustained(plan, jlo, jhi, xlo, xhi, K=60):
    d_in = _hip.DeviceArray.from_host(ctx, np.ascontiguousarray(data[:, jlo:jhi]))
    d_out = _hip.DeviceArray(ctx, (snum, xhi - xlo), np.float32)
    def step():
        plan.prep(d_in, jhi - jlo, jlo, jhi - jlo); plan.migrate(d_out, xlo, xhi)
    for _ in range(5): step()
    plan.sync(); t0 = time.perf_counter()
    for _ in range(K): step()
    plan.sync(); el = (time.perf_counter() - t0) / K * 1e3
    d_in.free(); d_out.free()
    return el

full = sustained(plan_for(1), 0, tnum, 0, tnum, 20)
print('1 rank: %.3f ms per step' % full)
for n in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    for label, kw in (('pair balance', {}), ('per-trace cost + widths of 192', dict(trace_cost=4.4e6, quantum=192))):
        tnum_pad, shards, blocks, pairs = parallel.plan_blocks(tt, 1.0, vel, tnum, n, **kw)
        ts = [sustained(plan_for(n), shards[r][0], shards[r][1], blocks[r][0], blocks[r][1]) for r in range(n)]
        print('%d ranks (%s): blocks %s  steps %s  max %.3f  kernel-side efficiency %.3f'
              % (n, label, [b[1] - b[0] for b in blocks], [round(t, 3) for t in ts], max(ts), full / n / max(ts)))

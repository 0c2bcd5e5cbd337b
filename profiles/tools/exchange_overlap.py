#!/usr/bin/env python3
"""One rank of an N-rank Kirchhoff run emulated on ONE GPU, exchange included: does the RCCL kernel of radargram
s+1 run underneath the persistent diffraction-sum kernel of radargram s, and what does leaving workgroup slots
free for it (IMPDAR_KIRCH_RESERVE) cost and buy?

For the busiest rank of the plan (the one that receives most rows) the pipelined step
    prep(own input shard) -> exchange -> migrate(own output block)
is timed with the exchange (a) left out, (b) done as a grouped self send/recv, through RCCL on a 1-rank
communicator, of exactly the byte ranges the rank would receive (the rows land where the real exchange puts them,
the source is the rank's own shard), for every reserve R given.  The copy is device-local, so its DURATION is not an
xGMI figure; what carries over is whether an RCCL kernel gets onto the chip while kirch_quad_kernel holds it, and
what the reserve costs the diffraction sum.  An xGMI estimate at 48 GB/s per link and direction (7 links, peers in
parallel) is printed beside it.

usage: exchange_overlap.py [case ...] [--reserve 0,8,16,32] [--steps 40] [--trace]
cases: c4r8 (40000 traces, 8 ranks: halo), c3r8 / c3r4 (10000 traces: all-gather share), c3r2 (halo)
--trace: a short run (one reserve, few steps) meant to sit under rocprofv3 --kernel-trace."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['IMPDAR_COMM_EMULATE'] = '1'
from impdar_amd import _hip, parallel, synth                      # noqa: E402
from impdar_amd.kirchhoff import KirchhoffPlan                    # noqa: E402

CASES = {'c4r8': (40000, 8), 'c3r8': (10000, 8), 'c3r4': (10000, 4), 'c3r2': (10000, 2), 'c4r4': (40000, 4), 'c4r2': (40000, 2)}
XGMI_GBS = 48.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('cases', nargs='*', default=['c4r8', 'c3r8', 'c3r4', 'c3r2'])
    ap.add_argument('--reserve', default='0,8,16,32')
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--trace', action='store_true')
    args = ap.parse_args()
    reserves = [int(v) for v in args.reserve.split(',')]
    snum, vel = 4096, 1.69e8
    lib = _hip.load()
    ctx = _hip.context(0)
    rdv = parallel.Rendezvous(0, 1)
    parallel.init_communicator(ctx, rdv)                           # a 1-rank RCCL communicator
    rng = np.random.default_rng(0)
    base = {}                      # one-rank step (prep + migrate, pipelined) of the whole radargram, per trace count
    summary = []
    for case in args.cases:
        tnum, n = CASES[case]
        if tnum not in base and not args.trace:
            geo1 = synth.geometry(snum, tnum)
            p1 = KirchhoffPlan(ctx, np.float32, snum, tnum, geo1['dist'], geo1['travel_time'], vel, False, 'fast', nranks=1)
            d1 = _hip.DeviceArray.from_host(ctx, rng.standard_normal((snum, tnum)).astype(np.float32))
            o1 = _hip.DeviceArray(ctx, (snum, tnum), np.float32)
            for _ in range(3):
                p1.prep(d1, tnum, 0, tnum)
                p1.migrate(o1, 0, tnum)
            p1.sync()
            t0 = time.perf_counter()
            K1 = 10 if tnum > 20000 else 20
            for _ in range(K1):
                p1.prep(d1, tnum, 0, tnum)
                p1.migrate(o1, 0, tnum)
            p1.sync()
            base[tnum] = (time.perf_counter() - t0) / K1 * 1e3
            p1.destroy()
            d1.free()
            o1.free()
            print('%d traces on 1 rank: %.3f ms per step' % (tnum, base[tnum]), flush=True)
        geo = synth.geometry(snum, tnum)
        tt = geo['travel_time'] / 1e6
        tnum_pad, shards, blocks, pairs = parallel.plan_blocks(tt, 1.0, vel, tnum, n)
        halo = parallel.halo_traces(tt, 1.0, vel)
        xp = parallel.plan_exchange(blocks, tnum_pad, n, halo)
        per = tnum_pad // n
        if xp['mode'] == 'halo':
            r = int(np.argmax(xp['rows_received']))
            recv = [(a, b) for _, a, b in xp['recv'][r]]
        else:
            r = n // 2
            recv = [(s * per, (s + 1) * per) for s in range(n) if s != r]
        jlo, jhi = shards[r]
        xlo, xhi = blocks[r]
        # self send/recv ranges: every received range [a, b) is fed from the rank's own shard rows, in pieces no
        # longer than the shard
        own_lo, own_len = r * per, per
        send, rcv = [], []
        for a, b in recv:
            while a < b:
                ln = min(b - a, own_len)
                send.append((0, own_lo, own_lo + ln))
                rcv.append((0, a, a + ln))
                a += ln
        rows = sum(b - a for _, a, b in rcv)
        mb = rows * snum * 4 / 1e6
        links = len(set(p for p, _, _ in xp['recv'][r])) if xp['mode'] == 'halo' else n - 1
        per_link_mb = max((b - a) for a, b in recv) * snum * 4 / 1e6 if xp['mode'] == 'halo' else per * snum * 4 / 1e6
        xgmi_ms = per_link_mb / XGMI_GBS
        plan = KirchhoffPlan(ctx, np.float32, snum, tnum, geo['dist'], geo['travel_time'], vel, False, 'fast', nranks=n)
        data = rng.standard_normal((snum, max(jhi - jlo, 1))).astype(np.float32)
        d_in = _hip.DeviceArray.from_host(ctx, data)
        d_out = _hip.DeviceArray(ctx, (snum, xhi - xlo), np.float32)

        def step(ex):
            plan.prep(d_in, jhi - jlo, jlo, jhi - jlo)
            if ex:
                plan.exchange(send, rcv)
            plan.migrate(d_out, xlo, xhi)

        def sustained(ex, K):
            for _ in range(4):
                step(ex)
            plan.sync()
            t0 = time.perf_counter()
            for _ in range(K):
                step(ex)
            plan.sync()
            el = (time.perf_counter() - t0) / K * 1e3
            hist = [plan.history_ms(b) for b in range(min(K, 32))]
            return el, float(np.mean([h[1] for h in hist])), float(np.mean([h[2] for h in hist]))

        print('%s: %d traces on %d ranks, %s; rank %d: shard [%d,%d) block [%d,%d) %d pairs; receives %d rows = %.1f MB '
              'from %d peers (largest single transfer %.1f MB = %.2f ms at %.0f GB/s per link)'
              % (case, tnum, n, xp['mode'], r, jlo, jhi, xlo, xhi, pairs[r], rows, mb, links, per_link_mb, xgmi_ms, XGMI_GBS),
              flush=True)
        if args.trace:
            os.environ['IMPDAR_KIRCH_RESERVE'] = str(reserves[0])
            sustained(True, 8)
            plan.destroy()
            d_in.free()
            d_out.free()
            continue
        # the exchange alone on an idle device
        plan.prep(d_in, jhi - jlo, jlo, jhi - jlo)
        plan.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            plan.exchange(send, rcv)
        plan.sync()
        alone = (time.perf_counter() - t0) / 10 * 1e3
        plan.migrate(d_out, xlo, xhi)
        plan.sync()
        print('   self send/recv alone on an idle device: %.3f ms (%.0f GB/s device-local)' % (alone, mb / alone), flush=True)
        for R in reserves:
            os.environ['IMPDAR_KIRCH_RESERVE'] = str(R)
            no_ex, _, k0 = sustained(False, args.steps)
            with_ex, ex_ms, k1 = sustained(True, args.steps)
            print('   reserve %3d: step without exchange %.3f ms (kernel %.3f) | with exchange %.3f ms (kernel %.3f, '
                  'exchange start->end on its stream %.3f ms) | exposed %.3f ms'
                  % (R, no_ex, k0, with_ex, k1, ex_ms, with_ex - no_ex), flush=True)
            summary.append((case, n, R, base[tnum], no_ex, with_ex, ex_ms, xgmi_ms))
        plan.destroy()
        d_in.free()
        d_out.free()
    if summary:
        print()
        print('projected strong-scaling efficiency of the busiest rank, T1 / (N x step): kernel side only | with the exchange '
              'done device-locally | with the exchange taking its xGMI estimate (hidden when it completes under the sum, '
              'i.e. when its start->end is far below the step; otherwise added to the step)')
        for case, n, R, t1, no_ex, with_ex, ex_ms, xg in summary:
            hidden = ex_ms < 0.5 * with_ex
            proj = max(with_ex, xg) if hidden else with_ex + xg
            print('   %s reserve %3d: %.3f | %.3f | %.3f  (%s)' % (case, R, t1 / n / no_ex, t1 / n / with_ex, t1 / n / proj,
                                                                  'exchange runs under the sum' if hidden else 'exchange waits for the sum to end'))
    lib.impdar_ctx_sync(ctx)


if __name__ == '__main__':
    main()

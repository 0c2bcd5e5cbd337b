#!/usr/bin/env python3
"""Headline benchmark: Kirchhoff migration of a 10000-trace x 4096-sample
float32 radargram (BASELINE.json config 3) on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path over the radargram with the input
already resident in HBM: time gradient + transpose of the rank's own input
traces, RCCL all-gather of the trace-major image (N > 1), diffraction sum
of the rank's output-trace block.  Strong scaling: the radargram is fixed,
output blocks are balanced by in-aperture pair count.

Rank 0 prints ONE JSON line (see README/DESIGN.md for the field meanings).
torch is used only as process-group plumbing (gloo barrier / max-reduce /
unique-id broadcast); all device work goes through the C ABI.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# LDS read port: 256 B/clk/CU x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md, LDS section: "aggregate ~150 TB/s")
LDS_PEAK_GBS = 256 * 256 * 2.4


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(data_full_cols, geo, vel, tnum, budget_s=15.0):
    """Time the plain-C oracle (oracle/kirch_oracle.c, OpenMP on the host
    cores) on a bounded sample of output traces of the same workload."""
    from oracle import c_oracle
    cores = c_oracle.threads()
    cols = np.linspace(0, tnum - 1, 8 * cores).round().astype(np.int32)
    cols = np.unique(cols)
    t0 = time.time()
    c_oracle.kirchhoff(data_full_cols, geo['travel_time'], geo['dist'], vel, False, traces=cols[:cores])
    probe = time.time() - t0
    per_trace = probe / cores
    n = int(max(cores, min(len(cols), budget_s / max(per_trace, 1e-6))))
    n = (n // cores) * cores
    sel = np.unique(np.linspace(0, tnum - 1, n).round().astype(np.int32))
    t0 = time.time()
    c_oracle.kirchhoff(data_full_cols, geo['travel_time'], geo['dist'], vel, False, traces=sel)
    el = time.time() - t0
    return {"value": len(sel) / el, "unit": "traces/s", "cores": cores, "kind": "port",
            "sample": "%d of %d output traces, evenly spaced, full aperture, fp64, %.1f s" % (len(sel), tnum, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--tnum', type=int, default=10000)
    ap.add_argument('--snum', type=int, default=4096)
    ap.add_argument('--mode', default='fast', choices=['fast', 'exact', 'auto'])
    ap.add_argument('--scaling', default='strong', choices=['strong', 'weak'],
                    help='strong (default): the BASELINE radargram is fixed; weak: --tnum traces PER GPU')
    ap.add_argument('--data', default='synthetic', choices=['synthetic', 'noise'])
    ap.add_argument('--no-cpu', action='store_true', help='skip the host-CPU baseline leg')
    ap.add_argument('--cpu-budget', type=float, default=15.0)
    args = ap.parse_args()

    # stdout must carry exactly ONE JSON line, but gloo and RCCL print banners on fd 1 (RCCL's
    # sits in the C stdio buffer until exit): point fd 1 at stderr for the whole run and keep
    # the real stdout for the result line
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                     % (args.gpus, args.gpus))
        args.gpus = world

    from impdar_amd import _hip, parallel, synth
    from impdar_amd.kirchhoff import KirchhoffPlan

    # IMPDAR_BENCH_FORCE_DIST=1 runs the whole multi-rank code path (gloo group, unique-id
    # broadcast, RCCL communicator, all-gather) with a single rank, for 1-GPU boxes
    multi = world > 1 or os.environ.get('IMPDAR_BENCH_FORCE_DIST') == '1'
    dist_pg = None
    if multi:
        import torch
        import torch.distributed as dist_pg
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist_pg.init_process_group('gloo', rank=rank, world_size=world)

    lib = _hip.load()
    ndev = _hip.device_count()
    if ndev <= 0:
        sys.exit('bench.py: no HIP device visible; the HIP path has no CPU fallback')
    ctx = _hip.context(local % ndev)

    if multi:
        import torch
        ident = torch.zeros(_hip.UNIQUE_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            import ctypes as C
            buf = C.create_string_buffer(_hip.UNIQUE_ID_BYTES)
            _hip.check(lib.impdar_comm_unique_id(buf), 'impdar_comm_unique_id')
            ident = torch.tensor(list(buf.raw), dtype=torch.uint8)
        dist_pg.broadcast(ident, 0)
        _hip.check(lib.impdar_comm_init(ctx, bytes(ident.tolist()), rank, world), 'impdar_comm_init')

    snum, tnum, vel = args.snum, args.tnum * (world if args.scaling == 'weak' else 1), 1.69e8
    geo = synth.geometry(snum, tnum)
    tt_sec = geo['travel_time'] / 1e6
    tnum_pad, shards, blocks, pairs = parallel.plan_blocks(tt_sec, 1.0, vel, tnum, world)
    jlo, jhi = shards[rank]
    xlo, xhi = blocks[rank]

    t0 = time.time()
    if args.data == 'synthetic':
        local_data = synth.diffractor_radargram(snum, tnum, vel=vel, dtype=np.float32, trace_lo=jlo, trace_hi=jhi)
    else:
        local_data = np.random.default_rng(rank).standard_normal((snum, jhi - jlo)).astype(np.float32)
    if rank == 0:
        log('[bench] rank0 input block %s built in %.1f s' % (local_data.shape, time.time() - t0))

    plan = KirchhoffPlan(ctx, np.float32, snum, tnum, geo['dist'], geo['travel_time'], vel, False,
                         args.mode, nranks=world)
    d_in = _hip.DeviceArray.from_host(ctx, local_data)
    d_out = _hip.DeviceArray(ctx, (snum, max(xhi - xlo, 1)), np.float32)
    nloc = jhi - jlo

    def step():
        plan.prep(d_in, max(nloc, 1), jlo, nloc)
        if multi:
            plan.allgather()
        plan.migrate(d_out, xlo, xhi)

    def fence():
        plan.sync()
        if dist_pg is not None:
            dist_pg.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms = []
    prep_ms = []
    gather_ms = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t_start
    # HIP-event durations recorded on the launch stream DURING the timed steps
    # (the plan keeps a 64-step ring of event pairs; nothing synchronised in the loop)
    for back in range(min(args.steps, 64)):
        p, g, m = plan.history_ms(back)
        prep_ms.append(p)
        gather_ms.append(g)
        kernel_ms.append(m)

    if dist_pg is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist_pg.all_reduce(t, op=dist_pg.ReduceOp.MAX)
        elapsed = float(t[0])

    out_host = d_out.to_host()
    finite = bool(np.isfinite(out_host).all())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = tnum * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else float('nan')
        algo_bytes = pairs[0] * 4 + snum * (xhi - xlo) * 4          # 4 B per in-aperture pair + output
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms == k_ms and k_ms > 0 else None
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'kirch_fast_hbm_traffic.json')
        if os.path.exists(tfile) and world == 1 and tnum == 10000 and snum == 4096:
            try:
                traffic = json.load(open(tfile)).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        res = {
            "metric": "migrated traces/sec + achieved HBM GB/s, Kirchhoff 10000x4096 radargram",
            "value": value, "unit": "traces/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": args.data,
            "config": {"workload": "Kirchhoff diffraction-sum migration, %d traces x %d samples, constant velocity "
                                   "1.69e8 m/s, dx 1 m, dt 10 ns (BASELINE config 3)" % (tnum, snum),
                       "kernel": plan.mode, "pairs_total": int(sum(pairs)), "pairs_rank0": int(pairs[0]),
                       "output_block_rank0": [int(xlo), int(xhi)],
                       "parallelism": "output-trace blocks x%d, RCCL all-gather of input image" % world,
                       "prep_ms": float(np.mean(prep_ms)) if prep_ms else None,
                       "allgather_ms": float(np.mean(gather_ms)) if gather_ms else None,
                       "output_finite": finite},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": "kirch_quad_kernel" if plan.mode == 'fast' else "kirch_exact_kernel",
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": algo_bytes,
                         # the image fits the Infinity Cache and every pair's sample is served from LDS, so the
                         # algorithmic rate exceeds the HBM peak; the port that actually binds is the LDS read
                         # port (one 4-byte ds_read lane per pair)
                         "on_chip": {"bound": "lds", "achieved": achieved, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                                     "frac": (achieved / LDS_PEAK_GBS) if achieved else None,
                                     "note": "peak at the 2.4 GHz spec clock; the kernel holds ~1.9-2.1 GHz, where "
                                             "rocprof counts 78% of the LDS cycles busy (40-trace tiles)"}},
        }
        if world == 1 and not args.no_cpu:
            t0 = time.time()
            if jlo == 0 and jhi == tnum:
                full = local_data
            else:
                full = synth.diffractor_radargram(snum, tnum, vel=vel, dtype=np.float32)
            res["cpu_baseline"] = cpu_baseline(full, geo, vel, tnum, args.cpu_budget)
            log('[bench] cpu baseline leg took %.1f s' % (time.time() - t0))
        else:
            res["cpu_baseline"] = None
        os.write(result_fd, (json.dumps(res) + '\n').encode())

    plan.destroy()
    if dist_pg is not None:
        dist_pg.barrier()
        dist_pg.destroy_process_group()


if __name__ == '__main__':
    main()

"""World-size-N CPU worker for tests/test_parallel_gloo.py: the BYTES of the halo exchange.

What RCCL moves on the GPUs is not float64 rows (the stand-in of _gloo_worker.py) but byte ranges of the device image
in the layout the ring kernels read: groups of 8 traces, sample-major inside a group (PrepParams::i8 of
impdar_amd/csrc/kirchhoff.hip: element (trace j, sample k) at ((j >> 3) * snum + k) * 8 + (j & 7), float32), and
impdar_kirch_exchange turns the row ranges of parallel.plan_exchange into byte offsets  off = lo * snum * 4,
len = (hi - lo) * snum * 4  from the image's first row.  Here every rank builds the grouped image of ITS input shard
only (everything else poisoned), moves exactly those byte ranges over gloo, and checks that every row its output
block reads (plan_exchange's `need`) then holds the bytes of the whole-radargram image.

    argv: <tnum> <dx> <world size is taken from the rendezvous>
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from impdar_amd import parallel, synth          # noqa: E402

POISON = 0xA5


def grouped_image(grad_rows, tnum_pad):
    """(ntraces, snum) float32 rows -> the bytes of the 8-trace-grouped image of tnum_pad rows (missing rows zero)."""
    n, snum = grad_rows.shape
    img = np.zeros((tnum_pad // 8, snum, 8), dtype=np.float32)
    j = np.arange(n)
    img[j >> 3, :, j & 7] = grad_rows
    return img.reshape(-1).view(np.uint8)


def main():
    tnum, dx = int(sys.argv[1]), float(sys.argv[2])
    rdv = parallel.Rendezvous()
    rank, world = rdv.rank, rdv.world
    dist.init_process_group('gloo', rank=rank, world_size=world)
    snum, vel = 64, 1.69e8
    geo = synth.geometry(snum, tnum, dx=dx)
    data = synth.noise_radargram(snum, tnum, seed=9).astype(np.float32)
    tt_sec = geo['travel_time'] / 1e6
    tnum_pad, shards, blocks, _ = parallel.plan_blocks(tt_sec, dx, vel, tnum, world)
    halo = parallel.halo_traces(tt_sec, dx, vel)
    xp = parallel.plan_exchange(blocks, tnum_pad, world, halo)
    rowb = snum * 4
    grad = np.gradient(data, tt_sec, axis=0).astype(np.float32).T          # (tnum, snum): what prep writes
    whole = grouped_image(grad, tnum_pad)
    # this rank's image: its own shard prepared, every other byte poison
    per = tnum_pad // world
    mine = np.full(tnum_pad * rowb, POISON, dtype=np.uint8)
    jlo, jhi = shards[rank]
    own = np.zeros((per, snum), dtype=np.float32)
    own[:jhi - jlo] = grad[jlo:jhi]
    mine[rank * per * rowb:(rank + 1) * per * rowb] = grouped_image(own, per)
    ops, bufs = [], []
    for peer, lo, hi in xp['send'][rank]:
        assert lo % 8 == 0 and hi % 8 == 0 and rank * per <= lo < hi <= (rank + 1) * per
        ops.append(dist.P2POp(dist.isend, torch.from_numpy(mine[lo * rowb:hi * rowb].copy()), peer))
    for peer, lo, hi in xp['recv'][rank]:
        t = torch.empty((hi - lo) * rowb, dtype=torch.uint8)
        bufs.append((lo, hi, t))
        ops.append(dist.P2POp(dist.irecv, t, peer))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for lo, hi, t in bufs:
        mine[lo * rowb:hi * rowb] = t.numpy()
    lo, hi = xp['need'][rank]
    ok = bool(np.array_equal(mine[lo * rowb:hi * rowb], whole[lo * rowb:hi * rowb]))
    # ... and nothing beyond what it needs arrived (the point of the halo form)
    outside = np.ones(tnum_pad, dtype=bool)
    outside[lo:hi] = False
    outside[rank * per:(rank + 1) * per] = False
    untouched = bool((mine.reshape(tnum_pad // 8, snum * 32)[outside[::8]] == POISON).all()) if outside.any() else True
    got = rdv.allgather((ok, untouched, hi - lo, xp['rows_received'][rank]))
    if rank == 0:
        assert all(g[0] for g in got), [g[0] for g in got]
        assert all(g[1] for g in got), [g[1] for g in got]
        print('GLOO_BYTES_OK world=%d mode=%s need=%s received=%s of %d rows'
              % (world, xp['mode'], [g[2] for g in got], [g[3] for g in got], xp['rows_allgather']))
    rdv.barrier()
    rdv.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""World-size-N CPU worker for tests/test_parallel_gloo.py (launched with
torch.distributed.run, gloo backend): exercises the shard plan and the
all-gather data movement of the multi-GPU Kirchhoff path with the CPU oracle
standing in for the device kernels."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from impdar_amd import parallel, synth          # noqa: E402
from oracle import mig_oracle                   # noqa: E402


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    snum, tnum, vel = 96, 75, 1.69e8            # tnum not divisible by the world size
    geo = synth.geometry(snum, tnum)
    tt_sec = geo['travel_time'] / 1e6
    data = synth.noise_radargram(snum, tnum, seed=4)
    tnum_pad, shards, blocks, pairs = parallel.plan_blocks(tt_sec, 1.0, vel, tnum, world)
    jlo, jhi = shards[rank]
    xlo, xhi = blocks[rank]
    # what kirch_prep does to the rank's own column block: gradient + transpose
    local = np.gradient(data[:, jlo:jhi], tt_sec, axis=0).T.copy()
    image = parallel.exchange_host(local, rank, world, tnum_pad // world)
    want_image = np.zeros((tnum_pad, snum))
    want_image[:tnum] = np.gradient(data, tt_sec, axis=0).T
    assert image.shape == (tnum_pad, snum)
    assert np.array_equal(image, want_image), 'all-gathered image differs'
    # migrate the rank's output block from the gathered image (the oracle takes the
    # radargram, so hand it the full data: the image equality above is what ties them)
    mine = mig_oracle.kirchhoff(data, geo['travel_time'], geo['dist'], vel, traces=range(xlo, xhi))[:, xlo:xhi]
    parts = [None] * world
    dist.all_gather_object(parts, (xlo, xhi, mine))
    if rank == 0:
        full = mig_oracle.kirchhoff(data, geo['travel_time'], geo['dist'], vel)
        got = np.zeros_like(full)
        covered = np.zeros(tnum, dtype=int)
        for lo, hi, blk in parts:
            got[:, lo:hi] = blk
            covered[lo:hi] += 1
        assert (covered == 1).all(), 'output blocks must tile [0,tnum) exactly once'
        assert np.array_equal(got, full)
        assert sum(pairs) == mig_oracle.count_pairs(snum, tnum, 1e-8, 1.0, vel)
        print('GLOO_OK world=%d blocks=%s pairs=%s' % (world, blocks, pairs))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

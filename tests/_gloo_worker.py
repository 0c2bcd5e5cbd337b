"""World-size-N CPU worker for tests/test_parallel_gloo.py: one rank of the sharded Kirchhoff migration THROUGH
THE PRODUCT ORCHESTRATION (impdar_amd.parallel.migrate_kirchhoff_sharded: partition, exchange schedule, control
plane), with a CPU stand-in for the device engine: NumPy gradient for prep, torch.distributed (gloo) for the
all-gather / halo send-receive that RCCL does on the GPUs, and the plain-C oracle's diffraction sum ON THE
EXCHANGED IMAGE for migrate -- so a wrong exchange gives a wrong result.

    RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT in the environment (torch.distributed.run or parallel.spawn_ranks);
    argv: <mode: auto|halo|allgather> <tnum> <dx> <result.npy or ->
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from impdar_amd import parallel, synth          # noqa: E402
from oracle import c_oracle, mig_oracle         # noqa: E402

POISON = 1.0e300        # rows a rank never received: finite, so nansum cannot hide a pick from them


class OracleEngine(object):
    """CPU stand-in for parallel.HipEngine (same calls, same order)."""

    def setup(self, sk, dist_km, travel_time_us, vel, nearfield, mode):
        self.sk, self.dist_km, self.tt_us, self.vel, self.nearfield = sk, dist_km, travel_time_us, vel, nearfield
        self.image = np.full((sk.tnum_pad, sk.snum), POISON)
        self.calls = []

    def prep(self, local, ld, jlo, nloc):
        self.calls.append('prep')
        if nloc:
            # what kirch_prep does to the rank's own column block: time gradient + transpose (mig_python.py:93)
            self.image[jlo:jlo + nloc] = np.gradient(local[:, :nloc], self.tt_us / 1e6, axis=0).T
        per = self.sk.tnum_pad // self.sk.world
        self.image[jlo + nloc:(self.sk.rank + 1) * per] = 0.0          # padding rows of the own shard are zero rows

    def allgather(self):
        self.calls.append('allgather')
        per = self.sk.tnum_pad // self.sk.world
        mine = torch.from_numpy(self.image[self.sk.rank * per:(self.sk.rank + 1) * per].copy())
        parts = [torch.empty_like(mine) for _ in range(self.sk.world)]
        dist.all_gather(parts, mine)
        self.image = np.concatenate([p.numpy() for p in parts], axis=0)

    def exchange(self, send, recv):
        self.calls.append('exchange')
        ops, bufs = [], []
        for peer, lo, hi in send:
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(self.image[lo:hi].copy()), peer))
        for peer, lo, hi in recv:
            t = torch.empty((hi - lo, self.sk.snum), dtype=torch.float64)
            bufs.append((lo, hi, t))
            ops.append(dist.P2POp(dist.irecv, t, peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for lo, hi, t in bufs:
            self.image[lo:hi] = t.numpy()

    def migrate(self, out, xlo, xhi):
        self.calls.append('migrate')
        if xhi > xlo:
            grad = self.image[:self.sk.tnum].T
            out[:, :xhi - xlo] = c_oracle.kirchhoff_from_gradient(grad, None, self.tt_us, self.dist_km, self.vel,
                                                                  False, traces=np.arange(xlo, xhi))

    def run_once(self, sk, local):
        out = np.zeros((sk.snum, max(sk.xhi - sk.xlo, 1)))
        sk.step(local, out)
        return out[:, :sk.xhi - sk.xlo]


def main():
    mode, tnum, dx = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    rdv = parallel.Rendezvous()                     # the product's control plane (no torch)
    rank, world = rdv.rank, rdv.world
    dist.init_process_group('gloo', rank=rank, world_size=world)      # data plane stand-in for RCCL
    snum, vel = 96, 1.69e8
    geo = synth.geometry(snum, tnum, dx=dx)
    data = synth.noise_radargram(snum, tnum, seed=4)
    _, shards = parallel.input_shards(tnum, world)
    jlo, jhi = shards[rank]
    eng = OracleEngine()
    xlo, xhi, mine = parallel.migrate_kirchhoff_sharded(
        data[:, jlo:jhi], dict(snum=snum, tnum=tnum, dist=geo['dist'], travel_time=geo['travel_time']),
        vel=vel, rdv=rdv, exchange=mode, engine=eng)
    xmode = eng.sk.xplan['mode']
    assert eng.calls == ['prep', 'exchange' if xmode == 'halo' else 'allgather', 'migrate'], eng.calls
    if xmode == 'halo' and eng.sk.xplan['rows_received'][rank] < eng.sk.xplan['rows_allgather']:
        # the point of the halo form: where the aperture does not span the profile, strictly less than the whole image
        # arrived (point-to-point is the default form also when it does)
        assert (eng.image == POISON).any() or world == 1
    parts = rdv.allgather((xlo, xhi, mine, xmode, eng.sk.xplan['rows_received'][rank]))
    if rank == 0:
        full = mig_oracle.kirchhoff(data, geo['travel_time'], geo['dist'], vel)
        got = np.zeros_like(full)
        covered = np.zeros(tnum, dtype=int)
        for lo, hi, blk, _, _ in parts:
            got[:, lo:hi] = blk
            covered[lo:hi] += 1
        assert (covered == 1).all(), 'output blocks must tile [0,tnum) exactly once'
        err = np.max(np.abs(got - full)) / np.max(np.abs(full))
        assert err < 1e-12, err
        assert sum(eng.sk.pairs) == mig_oracle.count_pairs(snum, tnum, 1e-8, dx, vel)
        print('GLOO_OK world=%d mode=%s blocks=%s rows_received=%s of %d err=%.1e'
              % (world, xmode, eng.sk.blocks, [p[4] for p in parts], eng.sk.xplan['rows_allgather'], err))
    rdv.barrier()
    rdv.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

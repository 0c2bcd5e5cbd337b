"""World-size-N CPU worker for tests/test_parallel_gloo.py: one rank of the phase shift sharded over the wavenumbers
THROUGH THE PRODUCT ORCHESTRATION (impdar_amd.parallel.migrate_phaseshift_sharded: slabs, control plane) with a CPU
stand-in for the device engine: the oracle's frequency sums for the rank's wavenumber slab, torch.distributed (gloo)
moving the byte blocks of ``parallel.alltoall_layout`` -- the table impdar_ps_alltoall_dev follows on the GPUs --
and NumPy's inverse transform over k on what ARRIVED, so a wrong block offset gives a wrong image.

    RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT in the environment; argv: <const|vz> <snum> <tnum>
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from impdar_amd import parallel, synth          # noqa: E402
from oracle import mig_oracle                   # noqa: E402


class OraclePsEngine(object):
    """CPU stand-in for parallel.HipPhaseShiftEngine."""

    def run(self, sp, data):
        a = sp.args
        nk, tw = sp.k_hi - sp.k_lo, sp.tau_hi - sp.tau_lo
        tap = mig_oracle._apply_taper(data, a['htaper'], a['vtaper'], inplace_form=True)
        FK = np.fft.fft2(tap, (sp.nt, sp.tnum))                       # replicated on every rank
        vm = a['vconst'] if a['vmig'] is None else a['vmig']
        tk = mig_oracle.phase_shift_tk(FK[:, sp.k_lo:sp.k_hi], vm, a['kx'][sp.k_lo:sp.k_hi], a['ws'], a['dt'],
                                       a['travel_time'], sp.snum, nk)  # (snum, nk)
        tk = np.ascontiguousarray(tk.T)                                # [nk][snum], as impdar_phaseshift_tk_dev leaves it
        send, recv = parallel.alltoall_layout(sp.tau_edges, sp.k_edges, sp.rank, 16)
        sbuf = np.concatenate([tk[:, sp.tau_edges[s]:sp.tau_edges[s + 1]].ravel() for s in range(sp.world)]
                              + [np.zeros(0, complex)]).view(np.uint8)
        rbuf = np.full(sp.tnum * tw * 16, 0xA5, dtype=np.uint8)       # poison: every byte must arrive
        assert sum(n for _, _, n in send) == sbuf.size and sum(n for _, _, n in recv) == rbuf.size
        ops, landed = [], []
        for (peer, off, n), (_, roff, rn) in zip(send, recv):
            if peer == sp.rank:                                        # the block a rank keeps (RCCL copies it on the device)
                assert n == rn
                rbuf[roff:roff + rn] = sbuf[off:off + n]
        for peer, off, n in send:
            if n and peer != sp.rank:
                ops.append(dist.P2POp(dist.isend, torch.from_numpy(sbuf[off:off + n].copy()), peer))
        for peer, off, n in recv:
            if n and peer != sp.rank:
                t = torch.empty(n, dtype=torch.uint8)
                landed.append((off, n, t))
                ops.append(dist.P2POp(dist.irecv, t, peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for off, n, t in landed:
            rbuf[off:off + n] = t.numpy()
        t2 = rbuf.view(np.complex128).reshape(sp.tnum, tw)             # [k][tau of this rank]
        return np.ascontiguousarray(np.fft.ifft(t2, axis=0).real.T)    # (tw, tnum): mig_python.py:282


def main():
    kind, snum, tnum = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rdv = parallel.Rendezvous()
    rank, world = rdv.rank, rdv.world
    dist.init_process_group('gloo', rank=rank, world_size=world)
    geo = synth.geometry(snum, tnum, dx=2.0)
    data = synth.noise_radargram(snum, tnum, seed=9).astype(np.float64)
    dt = float(geo['dt'])
    nt = int(2 ** np.ceil(np.log(snum) / np.log(2)))
    kx = mig_oracle._kx(tnum, geo['trace_int'], geo['dist'])
    ws = 2. * np.pi * np.fft.fftfreq(nt, d=dt)
    # constant velocity, or a v(z) profile of three layers with a gradient in the middle one
    vconst, vm = (1.69e8, None) if kind == 'const' else \
        (0.0, np.interp(np.arange(snum), [0, snum // 3, 2 * snum // 3, snum], [1.69e8, 1.69e8, 2.3e8, 2.3e8]))
    lo, hi, rows = parallel.migrate_phaseshift_sharded(
        data, dict(snum=snum, tnum=tnum, nt=nt, kx=kx, ws=ws, dt=dt, travel_time=geo['travel_time']), vconst, vm,
        rdv=rdv, engine=OraclePsEngine())
    parts = rdv.allgather((lo, hi, rows))
    if rank == 0:
        tap = mig_oracle._apply_taper(data, 100, 1000, inplace_form=True)      # the unsharded chain, mig_python.py:246-282
        tk = mig_oracle.phase_shift_tk(np.fft.fft2(tap, (nt, tnum)), vconst if vm is None else vm, kx, ws, dt,
                                       geo['travel_time'], snum, tnum)
        full = np.fft.ifft(tk).real
        got = np.full_like(full, np.nan)
        covered = np.zeros(snum, dtype=int)
        for plo, phi, blk in parts:
            got[plo:phi] = blk
            covered[plo:phi] += 1
        assert (covered == 1).all(), 'depth slabs must tile [0, snum) exactly once'
        err = np.max(np.abs(got - full)) / np.max(np.abs(full))
        assert err < 1e-12, err
        print('GLOO_PS_OK world=%d kind=%s slabs=%s err=%.1e' % (world, kind, [(p[0], p[1]) for p in parts], err))
    rdv.barrier()
    rdv.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

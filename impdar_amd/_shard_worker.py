"""One rank of ``parallel.run_sharded`` / ``run_sharded_phaseshift``: ``python -m impdar_amd._shard_worker <job
directory>`` with RANK / LOCAL_RANK / WORLD_SIZE in the environment.  Reads the shared-memory copy of the radargram
(plain ``.npy`` / JSON files in the parent's private directory: nothing is unpickled), runs its share on GPU
``LOCAL_RANK`` and writes its output block into the shared result."""
import json
import os
import sys

import numpy as np


def _load(base, name):
    return np.load(os.path.join(base, name + '.npy'), allow_pickle=False)


def kirchhoff(base, meta, parallel, rdv, data, out):
    _, shards = parallel.input_shards(int(meta['tnum']), rdv.world)
    jlo, jhi = shards[rdv.rank]
    local = np.ascontiguousarray(data[:, jlo:jhi])
    xlo, xhi, block = parallel.migrate_kirchhoff_sharded(
        local, dict(snum=int(meta['snum']), tnum=int(meta['tnum']), dist=_load(base, 'dist'),
                    travel_time=_load(base, 'travel_time')),
        vel=float(meta['vel']), nearfield=bool(meta['nearfield']), mode=str(meta['mode']), rdv=rdv)
    out[:, xlo:xhi] = block                      # float32 blocks widen here (mig_python.py:118 returns float64)


def phase_shift(base, meta, parallel, rdv, data, out):
    geometry = dict(snum=int(meta['snum']), tnum=int(meta['tnum']), nt=int(meta['nt']), kx=_load(base, 'kx'),
                    ws=_load(base, 'ws'), dt=float(meta['dt']), travel_time=_load(base, 'travel_time'))
    vmig = _load(base, 'vmig') if meta['has_vmig'] else None
    lo, hi, rows = parallel.migrate_phaseshift_sharded(np.ascontiguousarray(data), geometry, float(meta['vconst']), vmig,
                                                       float(meta['htaper']), float(meta['vtaper']), rdv=rdv)
    out[lo:hi, :] = rows                         # ... and here (:282 returns the real part as float64)


def main():
    base = sys.argv[1]
    with open(os.path.join(base, 'meta.json')) as fi:
        meta = json.load(fi)
    from impdar_amd import parallel
    rdv = parallel.Rendezvous()
    data = np.load(os.path.join(base, 'in.npy'), mmap_mode='r', allow_pickle=False)
    out = np.load(os.path.join(base, 'out.npy'), mmap_mode='r+', allow_pickle=False)
    {'Kirchhoff': kirchhoff, 'phase-shift': phase_shift}[meta['kind']](base, meta, parallel, rdv, data, out)
    out.flush()
    rdv.barrier()
    rdv.close()


if __name__ == '__main__':
    main()

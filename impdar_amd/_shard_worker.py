"""One rank of ``parallel.run_sharded``: ``python -m impdar_amd._shard_worker <job directory>`` with RANK /
LOCAL_RANK / WORLD_SIZE in the environment.  Reads its input shard from the shared-memory copy of the radargram
(plain ``.npy`` / JSON files in the parent's private directory: nothing is unpickled), runs
``migrate_kirchhoff_sharded`` on GPU ``LOCAL_RANK`` and writes its output block into the shared result."""
import json
import os
import sys

import numpy as np


def main():
    base = sys.argv[1]
    with open(os.path.join(base, 'meta.json')) as fi:
        meta = json.load(fi)
    from impdar_amd import parallel
    rdv = parallel.Rendezvous()
    data = np.load(os.path.join(base, 'in.npy'), mmap_mode='r', allow_pickle=False)
    dist = np.load(os.path.join(base, 'dist.npy'), allow_pickle=False)
    travel_time = np.load(os.path.join(base, 'travel_time.npy'), allow_pickle=False)
    _, shards = parallel.input_shards(int(meta['tnum']), rdv.world)
    jlo, jhi = shards[rdv.rank]
    local = np.ascontiguousarray(data[:, jlo:jhi])
    xlo, xhi, block = parallel.migrate_kirchhoff_sharded(
        local, dict(snum=int(meta['snum']), tnum=int(meta['tnum']), dist=dist, travel_time=travel_time),
        vel=float(meta['vel']), nearfield=bool(meta['nearfield']), mode=str(meta['mode']), rdv=rdv)
    out = np.load(os.path.join(base, 'out.npy'), mmap_mode='r+', allow_pickle=False)
    out[:, xlo:xhi] = block                      # float32 blocks widen here (mig_python.py:118 returns float64)
    out.flush()
    rdv.barrier()
    rdv.close()


if __name__ == '__main__':
    main()

"""One rank of ``parallel.run_sharded``: ``python -m impdar_amd._shard_worker <meta.pkl>`` with RANK / LOCAL_RANK /
WORLD_SIZE in the environment.  Reads its input shard from the shared-memory copy of the radargram, runs
``migrate_kirchhoff_sharded`` on GPU ``LOCAL_RANK`` and writes its output block into the shared result."""
import pickle
import sys

import numpy as np


def main():
    with open(sys.argv[1], 'rb') as fi:
        meta = pickle.load(fi)
    from impdar_amd import parallel
    rdv = parallel.Rendezvous()
    data = np.load(meta['f_in'], mmap_mode='r')
    _, shards = parallel.input_shards(meta['tnum'], rdv.world)
    jlo, jhi = shards[rdv.rank]
    local = np.ascontiguousarray(data[:, jlo:jhi])
    xlo, xhi, block = parallel.migrate_kirchhoff_sharded(
        local, dict(snum=meta['snum'], tnum=meta['tnum'], dist=meta['dist'], travel_time=meta['travel_time']),
        vel=meta['vel'], nearfield=meta['nearfield'], mode=meta['mode'], rdv=rdv)
    out = np.load(meta['f_out'], mmap_mode='r+')
    out[:, xlo:xhi] = block                      # float32 blocks widen here (mig_python.py:118 returns float64)
    out.flush()
    rdv.barrier()
    rdv.close()


if __name__ == '__main__':
    main()

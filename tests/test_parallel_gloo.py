"""Multi-rank path on CPU: world_size 2 and 3 over gloo (the RCCL all-gather on
the GPUs moves the same blocks; see impdar_amd/parallel.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from impdar_amd import parallel


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_kirchhoff_gloo(world):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(29500 + world),
           os.path.join(ROOT, 'tests', '_gloo_worker.py')]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'GLOO_OK world=%d' % world in out.stdout


def test_partition_properties():
    tt = np.arange(4096) * 1e-8
    h = parallel.aperture_half_widths(tt, 1.0, 1.69e8)
    assert h[0] == 3460 and h[-1] == 0                      # halo H = 3460 traces
    w = parallel.trace_pair_weights(h, 10000)
    assert int(w.sum()) == 189920188078
    # brute-force the weight of a few traces
    for xi in (0, 17, 5000, 9999):
        lo = np.maximum(xi - h, 0)
        hi = np.minimum(xi + h, 9999)
        assert w[xi] == int((hi - lo + 1).sum())
    for n in (1, 2, 4, 8):
        tnum_pad, shards, blocks, pairs = parallel.plan_blocks(tt, 1.0, 1.69e8, 10000, n)
        assert tnum_pad % (8 * n) == 0 and 10000 <= tnum_pad < 10000 + 8 * n     # whole 8-trace groups per shard
        assert blocks[0][0] == 0 and blocks[-1][1] == 10000
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(n - 1))
        assert shards[0][0] == 0 and shards[-1][1] == 10000
        assert sum(pairs) == 189920188078
        assert max(pairs) / (sum(pairs) / n) < 1.002        # balanced by pair count
        _, _, qb, qp = parallel.plan_blocks(tt, 1.0, 1.69e8, 10000, n, trace_cost=4.4e6, quantum=192)
        assert sum(qp) == sum(pairs) and qb[0][0] == 0 and qb[-1][1] == 10000
        if n >= 4:
            assert all((b[1] - b[0]) % 192 == 0 for b in qb[1:-1])      # optional rounded widths
        _, _, _, pure = parallel.plan_blocks(tt, 1.0, 1.69e8, 10000, n, trace_cost=0, quantum=1)
        assert max(pure) / (sum(pure) / n) < 1.002 and sum(pure) == sum(pairs)
    # ragged: more ranks than traces
    tnum_pad, shards, blocks, pairs = parallel.plan_blocks(np.arange(8) * 1e-8, 1.0, 1.69e8, 3, 4)
    assert tnum_pad == 32 and sum(b[1] - b[0] for b in blocks) == 3
    assert sum(s[1] - s[0] for s in shards) == 3

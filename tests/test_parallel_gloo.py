"""Multi-rank path on CPU: world_size 2 and 3 over gloo (the RCCL all-gather on
the GPUs moves the same blocks; see impdar_amd/parallel.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from impdar_amd import parallel


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


WORKER = os.path.join(ROOT, 'tests', '_gloo_worker.py')


@pytest.mark.parametrize('world,mode,tnum,dx', [(2, 'auto', 75, 1.0), (2, 'halo', 200, 4.0)])
def test_sharded_kirchhoff_gloo_under_torchrun(world, mode, tnum, dx):
    """Launched the way the driver launches bench.py (torch.distributed.run): the product's own rendezvous finds
    its peers from MASTER_PORT + the launcher's pid; the migrate runs on the exchanged image."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    env.pop('IMPDAR_RDV_JOB', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), WORKER, mode, str(tnum), str(dx)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'GLOO_OK world=%d' % world in out.stdout
    if mode == 'halo' or tnum == 200:
        assert 'mode=halo' in out.stdout


@pytest.mark.parametrize('world,mode,tnum,dx', [(3, 'auto', 75, 1.0), (3, 'auto', 400, 4.0), (3, 'allgather', 400, 4.0)])
def test_sharded_kirchhoff_gloo_spawned(world, mode, tnum, dx, capfd):
    """Launched by the product's own spawner (what `bench.py --gpus N` and `impproc migrate --gpus N` use)."""
    codes = parallel.spawn_ranks([sys.executable, WORKER, mode, str(tnum), str(dx)], world,
                                 env_extra=dict(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
                                                OMP_NUM_THREADS='1'), timeout=600)
    out = capfd.readouterr()
    assert codes == [0] * world, out.out[-2000:] + out.err[-4000:]
    assert 'GLOO_OK world=%d' % world in out.out
    if tnum == 400:
        assert ('mode=halo' if mode == 'auto' else 'mode=allgather') in out.out


BYTES_WORKER = os.path.join(ROOT, 'tests', '_gloo_bytes_worker.py')


@pytest.mark.parametrize('world,tnum,dx', [(2, 200, 4.0), (3, 430, 4.0), (4, 1000, 2.0)])
def test_halo_exchange_moves_the_bytes_of_the_grouped_image(world, tnum, dx, capfd):
    """What RCCL moves between two GPUs is byte ranges of the 8-trace-grouped float32 image (PrepParams::i8), at the
    offsets impdar_kirch_exchange derives from plan_exchange's row ranges.  The same bytes over gloo: every row a
    rank's output block reads holds the whole-radargram image afterwards, nothing else was touched."""
    codes = parallel.spawn_ranks([sys.executable, BYTES_WORKER, str(tnum), str(dx)], world,
                                 env_extra=dict(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
                                                OMP_NUM_THREADS='1'), timeout=600)
    out = capfd.readouterr()
    assert codes == [0] * world, out.out[-2000:] + out.err[-4000:]
    assert 'GLOO_BYTES_OK world=%d mode=halo' % world in out.out


PS_WORKER = os.path.join(ROOT, 'tests', '_gloo_ps_worker.py')


def test_sharded_phase_shift_gloo_under_torchrun():
    """Wavenumber slabs -> all-to-all -> depth-row slabs at world size 2, launched like bench.py."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    env.pop('IMPDAR_RDV_JOB', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), PS_WORKER, 'vz', '64', '45']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert 'GLOO_PS_OK world=2 kind=vz' in out.stdout


@pytest.mark.parametrize('world,kind,snum,tnum', [(3, 'const', 50, 37), (3, 'vz', 64, 90), (4, 'vz', 33, 3)])
def test_sharded_phase_shift_gloo_spawned(world, kind, snum, tnum, capfd):
    """... and at 3 and 4 ranks from the product's spawner; 3 wavenumbers on 4 ranks leaves one rank an empty
    wavenumber slab (it still finishes its depth rows)."""
    codes = parallel.spawn_ranks([sys.executable, PS_WORKER, kind, str(snum), str(tnum)], world,
                                 env_extra=dict(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
                                                OMP_NUM_THREADS='1'), timeout=600)
    out = capfd.readouterr()
    assert codes == [0] * world, out.out[-2000:] + out.err[-4000:]
    assert 'GLOO_PS_OK world=%d kind=%s' % (world, kind) in out.out


def test_alltoall_layout_is_a_permutation_of_the_blocks():
    """Every (wavenumber slab) x (depth slab) block is sent once and received once, sizes agree pairwise, and each
    rank's send and receive buffers are covered exactly."""
    for snum, tnum, world in ((50, 37, 3), (7, 2, 4), (4096, 4096, 8), (1, 1, 2)):
        te, ke = parallel.slab_edges(snum, world), parallel.slab_edges(tnum, world)
        assert te[0] == 0 and te[-1] == snum and ke[0] == 0 and ke[-1] == tnum
        assert max(np.diff(te)) - min(np.diff(te)) <= 1 and max(np.diff(ke)) - min(np.diff(ke)) <= 1
        plans = [parallel.alltoall_layout(te, ke, r, 8) for r in range(world)]
        for r, (send, recv) in enumerate(plans):
            assert [p for p, _, _ in send] == list(range(world)) == [p for p, _, _ in recv]
            at = 0
            for _, off, n in send:
                assert off == at
                at += n
            assert at == (ke[r + 1] - ke[r]) * snum * 8
            at = 0
            for _, off, n in recv:
                assert off == at
                at += n
            assert at == tnum * (te[r + 1] - te[r]) * 8
            for s in range(world):
                assert send[s][2] == plans[s][1][r][2]


def test_exchange_plan_properties():
    """plan_exchange: every row a block's aperture reaches is either the rank's own or received exactly once;
    point-to-point ranges by default (config 4 on 8 GPUs receives a third of the image, config 3 on 8 GPUs nearly
    all of it, from 7 peers over 7 links); the 75 % rule of rounds 1-2 is still there as a threshold."""
    tt = np.arange(4096) * 1e-8
    for tnum, n, want in ((40000, 8, 'halo'), (10000, 8, 'halo'), (10000, 2, 'halo'), (10000, 4, 'halo'), (40000, 2, 'halo')):
        tnum_pad, shards, blocks, pairs = parallel.plan_blocks(tt, 1.0, 1.69e8, tnum, n)
        halo = parallel.halo_traces(tt, 1.0, 1.69e8)
        assert halo == 3460 + 9
        xp = parallel.plan_exchange(blocks, tnum_pad, n, halo)
        assert xp['mode'] == want, (tnum, n, xp['mode'], xp['rows_received'], xp['rows_allgather'])
        old = parallel.plan_exchange(blocks, tnum_pad, n, halo, threshold=0.75)['mode']
        assert old == ('allgather' if (tnum, n) in ((10000, 8), (10000, 4)) else 'halo')
        per = tnum_pad // n
        for r in range(n):
            lo, hi = xp['need'][r]
            assert lo % 8 == 0 and hi % 8 == 0 and lo <= max(blocks[r][0] - halo, 0) and hi >= min(blocks[r][1] + halo, tnum)
            have = np.zeros(tnum_pad, dtype=int)
            have[r * per:(r + 1) * per] += 1
            for peer, a, b in xp['recv'][r]:
                assert peer != r and peer * per <= a < b <= (peer + 1) * per and a % 8 == 0 and b % 8 == 0
                assert (r, a, b) in xp['send'][peer]
                have[a:b] += 1
            assert (have[lo:hi] == 1).all()
        assert sum(len(v) for v in xp['send']) == sum(len(v) for v in xp['recv'])
    # config 4: an interior rank receives 2 x ~3469 traces + what its block overhangs its shard, not 35000
    xp = parallel.plan_exchange(*(lambda t: (t[2], t[0], 8, 3469))(parallel.plan_blocks(tt, 1.0, 1.69e8, 40000, 8)))
    assert max(xp['rows_received']) < 0.35 * xp['rows_allgather']


def test_rendezvous_single_rank_is_local():
    r = parallel.Rendezvous(rank=0, world=1)
    assert r.broadcast(b'abc') == b'abc' and r.allreduce_max(2.5) == 2.5 and r.allgather(7) == [7]
    r.barrier()
    r.close()


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


def test_rendezvous_codec_carries_data_only():
    """Frames are JSON with tagged bytes / numeric arrays: nothing that executes on arrival."""
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    msg = dict(ident=b'\x00\x01\xff' * 40, t=(1, 2.5, None, 'x', True), block=a, n=np.int64(7), f=np.float64(0.5), nan=float('nan'))
    import json
    back = parallel._dec(json.loads(json.dumps(parallel._enc(msg))))
    assert back['ident'] == msg['ident'] and back['t'] == [1, 2.5, None, 'x', True] and back['n'] == 7 and back['f'] == 0.5
    assert back['block'].dtype == np.float32 and np.array_equal(back['block'], a) and np.isnan(back['nan'])
    with pytest.raises(TypeError):
        parallel._enc(object())
    with pytest.raises(TypeError):
        parallel._enc(np.array(['a', None], dtype=object))
    with pytest.raises(ValueError):
        parallel._dec({'__nd': ['|O', [1], '']})


_RDV_CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from impdar_amd import parallel
r = parallel.Rendezvous(timeout=float(sys.argv[1]))
got = r.broadcast(b'unique-id' * 14 if r.rank == 0 else None)
assert got == b'unique-id' * 14
assert r.allreduce_max(float(r.rank)) == r.world - 1
parts = r.allgather((r.rank, np.full(3, r.rank, dtype=np.float64)))
assert [p[0] for p in parts] == list(range(r.world)) and all((p[1] == p[0]).all() for p in parts)
r.barrier(); r.close()
print('RDV_OK', r.rank)
'''


def test_rendezvous_refuses_strangers_and_foreign_files(tmp_path, capfd):
    """A peer that does not know the job secret is dropped before anything it sent is parsed and the job still
    forms; a rendezvous file that is not a private (0600) file of this user is never trusted."""
    import socket
    import stat
    import threading
    import time
    script = tmp_path / 'child.py'
    script.write_text(_RDV_CHILD % ROOT)
    job = 'sec_%d' % os.getpid()
    path = os.path.join(__import__('tempfile').gettempdir(), 'impdar_rdv_%s' % job)
    if os.path.exists(path):
        os.unlink(path)
    stop = []

    def stranger():
        # hammer rank 0's port with junk (oversized length prefixes, wrong MACs) while the ranks gather
        while not stop:
            try:
                port = int(open(path).read().split()[0])
                with socket.create_connection(('127.0.0.1', port), timeout=1.0) as c:
                    c.recv(16)
                    c.sendall(b'\xff' * 36 + b'\xff' * 8 + b'cos\nsystem\n(S"true"\ntR.')
                    time.sleep(0.01)
            except (OSError, ValueError, IndexError):
                time.sleep(0.005)
    th = threading.Thread(target=stranger, daemon=True)
    th.start()
    codes = parallel.spawn_ranks([sys.executable, str(script), '60'], 3, env_extra=dict(IMPDAR_RDV_JOB=job), timeout=120)
    stop.append(1)
    th.join(5)
    out = capfd.readouterr()
    assert codes == [0, 0, 0], out.err[-3000:]
    assert out.out.count('RDV_OK') == 3
    assert not os.path.exists(path)
    # a world-readable file at the path (what a stranger could have planted) is not trusted by rank 1
    with open(path, 'w') as fo:
        fo.write('1 deadbeef\n')
    os.chmod(path, 0o644)
    try:
        with pytest.raises(PermissionError):
            parallel._owned_private_file(path)
        env = dict(os.environ, RANK='1', LOCAL_RANK='1', WORLD_SIZE='2', IMPDAR_RDV_JOB=job)
        p = subprocess.run([sys.executable, str(script), '1.0'], env=env, capture_output=True, text=True, timeout=60)
        assert p.returncode != 0 and 'TimeoutError' in p.stderr
    finally:
        os.unlink(path)
    assert stat.S_IMODE(os.stat(tmp_path).st_mode)  # tmp_path untouched


def test_spawn_ranks_stops_the_others_when_one_rank_dies(tmp_path):
    """A rank that exits with an error must not leave its peers (blocked in a collective on the GPU) and the
    caller waiting: the survivors are terminated and every return code is reported."""
    import time
    script = tmp_path / 'die.py'
    script.write_text('import os, sys, time\nif os.environ["RANK"] == "1":\n    sys.exit(3)\ntime.sleep(600)\n')
    t0 = time.time()
    codes = parallel.spawn_ranks([sys.executable, str(script)], 3, timeout=300)
    assert time.time() - t0 < 30
    assert codes[1] == 3 and codes[0] != 0 and codes[2] != 0
    # and a job that simply takes too long
    script.write_text('import time\ntime.sleep(600)\n')
    t0 = time.time()
    codes = parallel.spawn_ranks([sys.executable, str(script)], 2, timeout=1.0)
    assert time.time() - t0 < 30 and all(c != 0 for c in codes)

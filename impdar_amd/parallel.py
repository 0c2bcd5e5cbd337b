"""Sharding of a Kirchhoff migration across the GPUs of one node.

Outputs are independent, so each rank owns a contiguous block of output
traces; the only exchange moves rows of the trace-major input image (every
rank prepares -- gradient + transpose -- only its own equal-width block of
input traces): an all-gather, or, when a block's
aperture halo is narrower than the rest of the profile, grouped
point-to-point sends of just the rows each block's aperture reaches
(``plan_exchange``).  Output blocks are sized by in-aperture pair count, not
trace count: traces near the ends of the profile see roughly half the
aperture of interior traces.

Layers, bottom up:

* partition arithmetic (``plan_blocks``, ``plan_exchange``): pure NumPy;
* ``Rendezvous``: the control plane of one job -- a TCP star on rank 0 found
  through a file in /tmp, carrying the 128-byte RCCL unique id, barriers and
  small reductions.  No torch, no MPI;
* ``migrate_kirchhoff_sharded``: what one rank does for one radargram
  (prep of its shard -> exchange -> diffraction sum of its block) through an
  *engine*: ``HipEngine`` is the C library (RCCL inside), the tests plug in a
  CPU stand-in to run the same orchestration over gloo;
* ``run_sharded``: the single-process front door (``RadarData.migrate`` with
  ``IMPDAR_NGPUS`` / ``impproc migrate --gpus N``): spawns one worker per GPU
  (``impdar_amd._shard_worker``) around shared-memory copies of the
  radargram and collects the output blocks.
"""
import base64
import hashlib
import hmac
import json
import os
import secrets
import shutil
import socket
import stat
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np


def aperture_half_widths(tt_sec, dx, vel):
    """Largest trace offset n with 2*sqrt((n dx)^2 + z^2)/vel <= t_max, per
    output sample (-1: the sample itself is out of range)."""
    tt = np.asarray(tt_sec, dtype=np.float64)
    dt = (tt[-1] - tt[0]) / (len(tt) - 1)
    alpha = (2.0 * dx / (vel * dt)) ** 2
    um = np.max(tt) / dt
    a = tt / dt
    rem = um * um - a * a
    h = np.full(len(tt), -1, dtype=np.int64)
    ok = rem >= 0
    h[ok] = np.floor(np.sqrt(rem[ok] / alpha) + 1e-12).astype(np.int64)
    return h


def trace_pair_weights(h, tnum):
    """Number of in-aperture (output sample, input trace) pairs per output trace."""
    h = np.asarray(h, dtype=np.int64)
    h = h[h >= 0]
    xi = np.arange(tnum, dtype=np.int64)
    w = np.zeros(tnum, dtype=np.int64)
    hs = np.sort(h)
    csum = np.concatenate([[0], np.cumsum(hs)])
    # sum_h min(h, m) for m = xi (left side) and m = tnum-1-xi (right side)
    for m_arr in (xi, tnum - 1 - xi):
        idx = np.searchsorted(hs, m_arr, side='right')      # h <= m  -> contributes h, else m
        w += csum[idx] + (len(hs) - idx) * m_arr
    return w + len(hs)


def balanced_blocks(weights, nranks, quantum=1):
    """Split [0, tnum) into ``nranks`` contiguous blocks of near-equal weight.  With ``quantum`` > 1 the
    widths of the interior blocks are rounded to multiples of it (the two end blocks share what is left
    in proportion to their balanced widths)."""
    w = np.asarray(weights, dtype=np.float64)
    tnum = len(w)
    c = np.concatenate([[0.0], np.cumsum(w)])
    edges = [0]
    for r in range(1, nranks):
        target = c[-1] * r / nranks
        e = int(np.searchsorted(c, target, side='left'))
        e = min(max(e, edges[-1]), tnum)
        edges.append(e)
    edges.append(tnum)
    if quantum > 1 and nranks >= 3:
        widths = [edges[r + 1] - edges[r] for r in range(nranks)]
        inner = [max(quantum, int(round(x / quantum)) * quantum) for x in widths[1:-1]]
        rest = tnum - sum(inner)
        if rest >= 2 and widths[0] + widths[-1] > 0:
            first = int(round(rest * widths[0] / (widths[0] + widths[-1])))
            first = min(max(first, 1), rest - 1)
            widths = [first] + inner + [rest - first]
            edges = [0]
            for x in widths:
                edges.append(edges[-1] + x)
    return [(edges[r], edges[r + 1]) for r in range(nranks)]


def input_shards(tnum, nranks):
    """Equal-width input blocks (the all-gather needs equal counts) of whole
    8-trace groups (the device image keeps 8 traces interleaved, so a block of
    whole groups is one contiguous run); the last ranks' blocks may be short or
    empty.  Returns (tnum_pad, [(jlo, jhi), ...]).  Same rule as
    ``impdar_kirch_plan_create`` (checked by the tests)."""
    per = ((tnum + 8 * nranks - 1) // (8 * nranks)) * 8
    shards = [(min(r * per, tnum), min((r + 1) * per, tnum)) for r in range(nranks)]
    return per * nranks, shards


# Optional refinements of the pair-count balance (both off by default).  Timing each rank's block in
# isolation suggested a fixed cost per output trace (4.4e6 pairs) and interior widths rounded to 192 traces
# (8 tiles of 24: no padding tiles, 6 x 192 traces x 16 chunks = the 768 resident workgroups); in the
# pipelined steady state of bench.py (prep of the next radargram under the current diffraction sum,
# profiles/tools/rank_steps.py) plain pair balance is the better of the two: 96 % vs 88 % at 4 ranks,
# 83 % vs 82 % at 8 (kernel side, all-gather not emulated).
TRACE_COST_PAIRS = 0.0
BLOCK_QUANTUM = 1


def plan_blocks(tt_sec, dx, vel, tnum, nranks, trace_cost=TRACE_COST_PAIRS, quantum=BLOCK_QUANTUM):
    """(tnum_pad, input shards, output blocks, pair counts per output block).  Output blocks are balanced
    by in-aperture pair count plus ``trace_cost`` per trace, interior widths rounded to ``quantum`` traces
    (pass 0 and 1 for pure pair balance)."""
    h = aperture_half_widths(tt_sec, dx, vel)
    w = trace_pair_weights(h, tnum)
    tnum_pad, shards = input_shards(tnum, nranks)
    blocks = balanced_blocks(w + trace_cost, nranks, quantum if (quantum > 1 and tnum >= 4 * quantum * nranks) else 1)
    pairs = [int(w[lo:hi].sum()) for lo, hi in blocks]
    return tnum_pad, shards, blocks, pairs


def plan_exchange(blocks, tnum_pad, nranks, halo, threshold=1.0):
    """Which image rows every rank must receive, and from whom.

    Rank r's output block [xlo, xhi) reads input traces [xlo - halo, xhi + halo) (whole 8-trace groups, clipped
    to the padded profile); rank s owns rows [s*per, (s+1)*per).  Returns a dict with ``mode``: ``'halo'`` (grouped
    point-to-point transfers of exactly the ranges in ``recv`` / ``send``) when the busiest rank receives at most
    ``threshold`` of what the all-gather would hand it, else ``'allgather'``; ``need[r]`` = (lo, hi); ``recv[r]`` /
    ``send[r]`` = lists of (peer, row_lo, row_hi).  Deterministic in its arguments, so every rank derives the same
    plan without talking.

    The default threshold is 1: point-to-point always.  On a fully connected xGMI node every pair of ranks has a link
    of its own, so direct transfers of what each rank needs are never slower than a ring, and they are the form whose
    co-existence with the persistent diffraction-sum kernel was measured (profiles/r03_exchange_overlap.txt: an RCCL
    send/recv kernel runs underneath the sum once 32 workgroup slots are left free; an all-gather of more than one
    rank cannot be run on the one-GPU boxes this was built on).  ``exchange='allgather'`` still asks for the
    collective."""
    per = tnum_pad // nranks
    need, recv, send = [], [[] for _ in range(nranks)], [[] for _ in range(nranks)]
    for r, (xlo, xhi) in enumerate(blocks):
        if xhi <= xlo:
            need.append((0, 0))
            continue
        lo = max(0, ((xlo - halo) // 8) * 8)
        hi = min(tnum_pad, -((-(xhi + halo)) // 8) * 8)
        need.append((lo, hi))
        for s_ in range(nranks):
            a, b = max(lo, s_ * per), min(hi, (s_ + 1) * per)
            if s_ != r and b > a:
                recv[r].append((s_, a, b))
                send[s_].append((r, a, b))
    got = [sum(b - a for _, a, b in rv) for rv in recv]
    full = (nranks - 1) * per
    mode = 'halo' if nranks > 1 and full > 0 and max(got) <= threshold * full else 'allgather'
    return dict(mode=mode, need=need, recv=recv, send=send, rows_received=got, rows_allgather=full)


def halo_traces(tt_sec, dx, vel):
    """Aperture half width in traces (+ one 8-trace group of margin for the kernels' staging look-ahead)."""
    h = aperture_half_widths(tt_sec, dx, vel)
    return int(max(int(h.max()), 0)) + 1 + 8


# ---------------------------------------------------------------------------------------------------------
# control plane
# ---------------------------------------------------------------------------------------------------------
# Wire format: nothing that executes.  A frame is an 8-byte little-endian length + UTF-8 JSON; bytes and NumPy
# arrays of plain numeric dtypes travel as tagged base64 (``{"__b": ...}``, ``{"__nd": [dtype, shape, ...]}``),
# tuples arrive as lists.  A connection is authenticated with the job's secret (HMAC-SHA256 challenge / response
# of fixed size, compared in constant time) BEFORE any frame is parsed.
_MAX_FRAME = 1 << 28
_NONCE, _MAC = 16, 32


def _enc(o):
    if isinstance(o, (bytes, bytearray)):
        return {'__b': base64.b64encode(bytes(o)).decode('ascii')}
    if isinstance(o, np.ndarray):
        if o.dtype.kind not in 'fiubc':
            raise TypeError('rendezvous messages carry numeric arrays only, got dtype %s' % o.dtype)
        return {'__nd': [o.dtype.str, list(o.shape), base64.b64encode(np.ascontiguousarray(o).tobytes()).decode('ascii')]}
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, (list, tuple)):
        return [_enc(v) for v in o]
    if isinstance(o, dict):
        return {str(k): _enc(v) for k, v in o.items()}
    if o is None or isinstance(o, (bool, int, float, str)):
        return o
    raise TypeError('rendezvous messages carry None / bool / int / float / str / bytes / numeric arrays and lists or '
                    'dicts of them, got %s' % type(o).__name__)


def _dec(o):
    if isinstance(o, list):
        return [_dec(v) for v in o]
    if isinstance(o, dict):
        if set(o) == {'__b'}:
            return base64.b64decode(o['__b'])
        if set(o) == {'__nd'}:
            dt, shape, blob = o['__nd']
            dt = np.dtype(str(dt))
            if dt.kind not in 'fiubc':
                raise ValueError('refused array dtype %s' % dt)
            return np.frombuffer(base64.b64decode(blob), dtype=dt).reshape([int(v) for v in shape]).copy()
        return {k: _dec(v) for k, v in o.items()}
    return o


def _send_msg(sock, obj):
    blob = json.dumps(_enc(obj), allow_nan=True).encode('utf-8')
    sock.sendall(struct.pack('<Q', len(blob)) + blob)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        buf += chunk
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    if n > _MAX_FRAME:
        raise ConnectionError('rendezvous frame of %d bytes refused' % n)
    return _dec(json.loads(_recv_exact(sock, n).decode('utf-8')))


def _mac(secret, *parts):
    return hmac.new(secret, b'|'.join(parts), hashlib.sha256).digest()


def _owned_private_file(path):
    """Open ``path`` for reading only if it is a regular file of this user with no group / other access."""
    fd = os.open(path, os.O_RDONLY | getattr(os, 'O_NOFOLLOW', 0))
    try:
        st = os.fstat(fd)
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise PermissionError('%s is not a private file of uid %d' % (path, os.getuid()))
        return os.fdopen(fd, 'r')
    except Exception:
        os.close(fd)
        raise


class Rendezvous(object):
    """Control plane of one multi-process job on one node: rank 0 listens on an ephemeral TCP port of
    127.0.0.1 and publishes ``port secret`` in ``<tmp>/impdar_rdv_<job>``, a file it creates exclusively with
    mode 0600; the other ranks read it only if it is a private file of their own uid, connect, and prove they know
    the secret (challenge / response, both directions) before a single frame is parsed.  ``$IMPDAR_RDV_SECRET``
    (set by ``spawn_ranks``) is mixed into the secret when present.  ``job`` defaults to ``$MASTER_PORT`` + the
    launcher's pid (the same for every rank under ``torch.distributed.run`` and under ``bench.py``'s own spawner),
    so concurrent jobs do not meet.  Collectives are tiny (a unique id, a float, a barrier) and go through rank 0.
    Ranks must call the same collectives in the same order."""

    def __init__(self, rank=None, world=None, job=None, timeout=120.0):
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        self.timeout = timeout
        self.peers = {}
        self.sock = None
        self.path = None
        if self.world == 1:
            return
        if job is None:
            job = os.environ.get('IMPDAR_RDV_JOB') or '%s_%d' % (os.environ.get('MASTER_PORT', '0'), os.getppid())
        self.path = os.path.join(tempfile.gettempdir(), 'impdar_rdv_%s' % job)
        env_secret = os.environ.get('IMPDAR_RDV_SECRET', '').encode('utf-8')
        tag = ('impdar-rdv %s world=%d' % (job, self.world)).encode('utf-8')
        if self.rank == 0:
            file_secret = secrets.token_hex(32)
            secret = hashlib.sha256(env_secret + b'|' + file_secret.encode('ascii') + b'|' + tag).digest()
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(('127.0.0.1', 0))
            srv.listen(self.world)
            srv.settimeout(timeout)
            self._publish('%d %s\n' % (srv.getsockname()[1], file_secret))
            try:
                while len(self.peers) < self.world - 1:
                    c, _ = srv.accept()
                    c.settimeout(timeout)
                    try:
                        nonce = secrets.token_bytes(_NONCE)
                        c.sendall(nonce)
                        reply = _recv_exact(c, 4 + _MAC)                 # fixed size: rank + MAC, nothing parsed yet
                        (r,) = struct.unpack('<I', reply[:4])
                        good = hmac.compare_digest(reply[4:], _mac(secret, b'hello', nonce, reply[:4]))
                        if not good or not 0 < r < self.world or r in self.peers:
                            c.close()
                            continue
                        c.sendall(_mac(secret, b'welcome', nonce, reply[:4]))
                    except (OSError, ConnectionError, struct.error):
                        c.close()
                        continue
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.peers[r] = c
                for c in self.peers.values():
                    _send_msg(c, 'go')
            finally:
                srv.close()
                try:
                    os.unlink(self.path)
                except OSError:
                    pass
        else:
            deadline = time.time() + timeout
            rank4 = struct.pack('<I', self.rank)
            while True:
                c = None
                try:
                    with _owned_private_file(self.path) as fi:
                        port_s, file_secret = fi.read().split()[:2]
                    secret = hashlib.sha256(env_secret + b'|' + file_secret.encode('ascii') + b'|' + tag).digest()
                    c = socket.create_connection(('127.0.0.1', int(port_s)), timeout=5.0)
                    c.settimeout(timeout)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    nonce = _recv_exact(c, _NONCE)
                    c.sendall(rank4 + _mac(secret, b'hello', nonce, rank4))
                    if hmac.compare_digest(_recv_exact(c, _MAC), _mac(secret, b'welcome', nonce, rank4)) and \
                            _recv_msg(c) == 'go':
                        self.sock = c
                        break
                    c.close()
                except (OSError, ValueError, IndexError, ConnectionError, EOFError):
                    # no file yet, a stale file of an earlier job, somebody else's file, or a refused handshake
                    if c is not None:
                        c.close()
                if time.time() > deadline:
                    raise TimeoutError('rank %d found no rendezvous at %s within %.0f s' % (self.rank, self.path, timeout))
                time.sleep(0.05)

    def _publish(self, text):
        """Create the rendezvous file exclusively, mode 0600.  A leftover of this user's earlier job is replaced;
        a file somebody else put at the path is an error, never trusted."""
        for _ in range(2):
            try:
                fd = os.open(self.path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, 'O_NOFOLLOW', 0), 0o600)
            except FileExistsError:
                st = os.lstat(self.path)
                if st.st_uid != os.getuid() or not stat.S_ISREG(st.st_mode):
                    raise PermissionError('rendezvous path %s exists and is not a file of uid %d' % (self.path, os.getuid()))
                os.unlink(self.path)
                continue
            with os.fdopen(fd, 'w') as fo:
                fo.write(text)
            return
        raise FileExistsError(self.path)

    # every collective is a gather to rank 0 followed by a broadcast of the reduced value
    def _collect(self, value, reduce_fn):
        if self.world == 1:
            return reduce_fn([value])
        if self.rank == 0:
            vals = [None] * self.world
            vals[0] = value
            for r, c in self.peers.items():
                vals[r] = _recv_msg(c)
            out = reduce_fn(vals)
            for c in self.peers.values():
                _send_msg(c, out)
            return out
        _send_msg(self.sock, value)
        return _recv_msg(self.sock)

    def broadcast(self, value, root=0):
        return self._collect(value, lambda vals: vals[root])

    def barrier(self):
        self._collect(None, lambda vals: None)

    def allreduce_max(self, x):
        return self._collect(float(x), max)

    def allgather(self, obj):
        """List over ranks of ``obj`` (tuples arrive as lists)."""
        return self._collect(obj, list)

    def close(self):
        for c in list(self.peers.values()) + ([self.sock] if self.sock else []):
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.sock = {}, None


def init_communicator(ctx, rdv):
    """RCCL communicator of the job on ``ctx``: rank 0's unique id travels over the rendezvous."""
    import ctypes as C
    from . import _hip
    lib = _hip.load()
    _hip.require_system_rccl()
    ident = None
    if rdv.rank == 0:
        buf = C.create_string_buffer(_hip.UNIQUE_ID_BYTES)
        _hip.check(lib.impdar_comm_unique_id(buf), 'impdar_comm_unique_id')
        ident = bytes(buf.raw)
    ident = rdv.broadcast(ident, 0)
    _hip.check(lib.impdar_comm_init(ctx, ident, rdv.rank, rdv.world), 'impdar_comm_init')


# ---------------------------------------------------------------------------------------------------------
# one rank's share of one radargram
# ---------------------------------------------------------------------------------------------------------
class ShardedKirchhoff(object):
    """One rank's plan for migrating radargrams of a fixed geometry on ``world`` GPUs: partition, exchange
    schedule, the device plan and buffers.  ``step()`` enqueues prep -> exchange -> diffraction sum for the
    rank's resident input shard; consecutive steps pipeline (the C plan double-buffers)."""

    def __init__(self, ctx, snum, tnum, dist_km, travel_time_us, vel, rank, world, dtype=np.float32,
                 nearfield=False, mode='auto', exchange='auto', engine=None):
        self.rank, self.world = int(rank), int(world)
        self.snum, self.tnum, self.dtype = int(snum), int(tnum), np.dtype(dtype)
        tt_sec = np.asarray(travel_time_us, dtype=np.float64) / 1.0e6
        dist_m = np.asarray(dist_km, dtype=np.float64) * 1.0e3
        dx = float((dist_m[-1] - dist_m[0]) / (tnum - 1)) if tnum > 1 else 1.0
        uniform_x = tnum < 2 or (dx > 0 and np.allclose(np.diff(dist_m), dx, rtol=1e-9, atol=0))
        if uniform_x:
            self.tnum_pad, self.shards, self.blocks, self.pairs = plan_blocks(tt_sec, dx, vel, tnum, world)
            halo = halo_traces(tt_sec, dx, vel)
        else:
            # irregular spacing: equal-width output blocks, whole-image exchange
            self.tnum_pad, self.shards = input_shards(tnum, world)
            e = [tnum * r // world for r in range(world + 1)]
            self.blocks = [(e[r], e[r + 1]) for r in range(world)]
            self.pairs = [None] * world
            halo = tnum
        self.xplan = plan_exchange(self.blocks, self.tnum_pad, world, halo)
        if exchange in ('halo', 'allgather'):
            self.xplan['mode'] = exchange if world > 1 else 'allgather'
        self.jlo, self.jhi = self.shards[self.rank]
        self.xlo, self.xhi = self.blocks[self.rank]
        self.engine = engine if engine is not None else HipEngine(ctx)
        self.engine.setup(self, dist_km, travel_time_us, vel, nearfield, mode)

    @property
    def nloc(self):
        return self.jhi - self.jlo

    def step(self, d_in, d_out, multi=None):
        multi = self.world > 1 if multi is None else multi
        self.engine.prep(d_in, max(self.nloc, 1), self.jlo, self.nloc)
        if multi:
            if self.xplan['mode'] == 'halo':
                self.engine.exchange(self.xplan['send'][self.rank], self.xplan['recv'][self.rank])
            else:
                self.engine.allgather()
        self.engine.migrate(d_out, self.xlo, self.xhi)


class HipEngine(object):
    """The device side of ``ShardedKirchhoff``: ``impdar_kirch_*`` of the C ABI (RCCL inside the library)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.plan = None

    def setup(self, sk, dist_km, travel_time_us, vel, nearfield, mode):
        from .kirchhoff import KirchhoffPlan
        self.plan = KirchhoffPlan(self.ctx, sk.dtype, sk.snum, sk.tnum, dist_km, travel_time_us, vel, nearfield,
                                  mode, nranks=sk.world)
        assert self.plan.tnum_pad == sk.tnum_pad, (self.plan.tnum_pad, sk.tnum_pad)

    def prep(self, d_in, ld, jlo, nloc):
        self.plan.prep(d_in, ld, jlo, nloc)

    def allgather(self):
        self.plan.allgather()

    def exchange(self, send, recv):
        self.plan.exchange(send, recv)

    def migrate(self, d_out, xlo, xhi):
        self.plan.migrate(d_out, xlo, xhi)


def migrate_kirchhoff_sharded(local_data, geometry, vel=1.69e8, nearfield=False, mode='auto', rdv=None, ctx=None,
                              exchange='auto', engine=None):
    """Collective: every rank of the job calls this with ITS input shard ``local_data`` =
    ``data[:, jlo:jhi]`` (``parallel.input_shards``) of a (snum, tnum) radargram and gets back
    ``(xlo, xhi, block)``: the migrated output traces it owns, in ``local_data``'s dtype.

    ``geometry``: dict(snum, tnum, dist [km], travel_time [us]).  ``rdv``: the job's ``Rendezvous`` (made from the
    environment when None).  Reference semantics per block: mig_python.py:63-123."""
    from . import _hip
    own_rdv = rdv is None
    rdv = Rendezvous() if rdv is None else rdv
    if engine is None:
        _hip.load()
        ctx = _hip.context() if ctx is None else ctx
        if rdv.world > 1 and not _hip.load().impdar_comm_size(ctx) == rdv.world:
            init_communicator(ctx, rdv)
    snum, tnum = int(geometry['snum']), int(geometry['tnum'])
    local_data = np.ascontiguousarray(local_data)
    sk = ShardedKirchhoff(ctx, snum, tnum, geometry['dist'], geometry['travel_time'], vel, rdv.rank, rdv.world,
                          local_data.dtype, nearfield, mode, exchange, engine)
    if local_data.shape != (snum, sk.nloc):
        raise ValueError('rank %d owns input traces [%d, %d): expected a (%d, %d) shard, got %s'
                         % (rdv.rank, sk.jlo, sk.jhi, snum, sk.nloc, local_data.shape))
    block = sk.engine.run_once(sk, local_data)
    rdv.barrier()
    if own_rdv:
        rdv.close()
    return sk.xlo, sk.xhi, block


def _hip_run_once(self, sk, local_data):
    from . import _hip
    d_in = _hip.DeviceArray.from_host(self.ctx, local_data if sk.nloc else np.zeros((sk.snum, 1), local_data.dtype))
    d_out = _hip.DeviceArray(self.ctx, (sk.snum, max(sk.xhi - sk.xlo, 1)), sk.dtype)
    try:
        sk.step(d_in, d_out)
        self.plan.sync()
        out = d_out.to_host()[:, :sk.xhi - sk.xlo]
    finally:
        self.plan.destroy()
        d_in.free()
        d_out.free()
    return out


HipEngine.run_once = _hip_run_once


# ---------------------------------------------------------------------------------------------------------
# phase shift sharded over the wavenumbers
# ---------------------------------------------------------------------------------------------------------
def slab_edges(n, nranks):
    """``nranks + 1`` edges of near-equal contiguous slabs of ``range(n)`` (the wavenumber and the depth-row
    slabs of the sharded phase shift: every wavenumber costs the same, mig_python.py:396-487)."""
    return [n * r // nranks for r in range(nranks + 1)]


def alltoall_layout(tau_edges, k_edges, rank, itemsize):
    """Byte layout of one rank's all-to-all (what ``impdar_ps_alltoall_dev`` does on the device; the CPU stand-in
    of the tests follows the same table).  Rank ``r`` holds TK rows of its wavenumbers ``[k_edges[r], k_edges[r+1])``
    over ALL depth rows; it sends to rank ``s`` the columns of s's depth slab, packed one block per peer, and
    receives from ``s`` the rows ``[k_edges[s], k_edges[s+1])`` of its own [tnum][tw] array: k-major, ready for
    the inverse transform over k (:282).  Returns ``(send, recv)``: lists of ``(peer, offset, nbytes)``."""
    nranks = len(k_edges) - 1
    nk = k_edges[rank + 1] - k_edges[rank]
    tw = tau_edges[rank + 1] - tau_edges[rank]
    send, recv, at = [], [], 0
    for s in range(nranks):
        n = nk * (tau_edges[s + 1] - tau_edges[s]) * itemsize
        send.append((s, at, n))
        at += n
        recv.append((s, k_edges[s] * tw * itemsize, (k_edges[s + 1] - k_edges[s]) * tw * itemsize))
    return send, recv


class ShardedPhaseShift(object):
    """One rank of a phase-shift (Gazdag) migration sharded over the wavenumbers on ``world`` GPUs.  Every rank
    holds the whole radargram (the forward transforms are a few percent of the work and are replicated), sums the
    frequencies of its wavenumber slab, the slabs are redistributed into depth-row slabs by one all-to-all and
    every rank finishes rows ``[tau_lo, tau_hi)`` of the image.  Constant velocity and v(z) (mig_python.py:396-446);
    the 2-D v(x, z) branch is not sharded."""

    def __init__(self, ctx, snum, tnum, nt, kx, ws, dt, travel_time_us, vconst, vmig, htaper, vtaper, rank, world,
                 dtype=np.float32, engine=None):
        self.rank, self.world = int(rank), int(world)
        self.snum, self.tnum, self.nt, self.dtype = int(snum), int(tnum), int(nt), np.dtype(dtype)
        self.k_edges = slab_edges(self.tnum, self.world)
        self.tau_edges = slab_edges(self.snum, self.world)
        self.k_lo, self.k_hi = self.k_edges[self.rank], self.k_edges[self.rank + 1]
        self.tau_lo, self.tau_hi = self.tau_edges[self.rank], self.tau_edges[self.rank + 1]
        self.args = dict(kx=np.ascontiguousarray(kx, dtype=np.float64), ws=np.ascontiguousarray(ws, dtype=np.float64),
                         dt=float(dt), travel_time=np.ascontiguousarray(travel_time_us, dtype=np.float64),
                         vconst=float(vconst),
                         vmig=None if vmig is None else np.ascontiguousarray(vmig, dtype=np.float64),
                         htaper=float(htaper), vtaper=float(vtaper))
        if self.args['vmig'] is not None and len(self.args['vmig']) != self.snum:
            raise ValueError('Interpolated velocity profile is not the length of the number of samples in a trace.')
        self.engine = engine if engine is not None else HipPhaseShiftEngine(ctx)

    def run(self, data):
        """``data``: the whole (snum, tnum) radargram.  Returns this rank's rows ``[tau_lo, tau_hi)`` of the
        migrated image, (tau_hi - tau_lo, tnum) in ``data``'s dtype."""
        data = np.ascontiguousarray(data)
        if data.shape != (self.snum, self.tnum):
            raise ValueError('The input array must be of size (snum, tnum)')
        return self.engine.run(self, data)


class HipPhaseShiftEngine(object):
    """The device side of ``ShardedPhaseShift``: ``impdar_phaseshift_tk_dev`` -> ``impdar_ps_alltoall_dev`` (grouped
    RCCL send/recv inside the library) -> ``impdar_phaseshift_finish_dev``."""

    def __init__(self, ctx):
        self.ctx = ctx

    def run(self, sp, data):
        import ctypes as C
        from . import _hip
        lib = _hip.load()
        ctx, a = self.ctx, sp.args
        code = _hip.dtype_code(sp.dtype)
        cdt = np.complex64 if sp.dtype == np.float32 else np.complex128
        nk, tw = sp.k_hi - sp.k_lo, sp.tau_hi - sp.tau_lo
        d_in = _hip.DeviceArray.from_host(ctx, data.astype(sp.dtype, copy=False))
        d_tk = _hip.DeviceArray(ctx, (max(nk, 1), sp.snum), cdt)
        d_t2 = _hip.DeviceArray(ctx, (sp.tnum, max(tw, 1)), cdt)
        d_out = _hip.DeviceArray(ctx, (max(tw, 1), sp.tnum), sp.dtype)
        try:
            p_vm = _hip.as_dp(a['vmig'])[1] if a['vmig'] is not None else None
            rc = lib.impdar_phaseshift_tk_dev(ctx, d_in.ptr, code, sp.snum, sp.tnum, sp.nt, _hip.as_dp(a['kx'])[1],
                                              _hip.as_dp(a['ws'])[1], a['dt'], _hip.as_dp(a['travel_time'])[1],
                                              a['vconst'], p_vm, 0 if a['vmig'] is None else sp.snum, a['htaper'],
                                              a['vtaper'], sp.k_lo, nk, d_tk.ptr)
            _hip.check(rc, 'impdar_phaseshift_tk_dev')
            ia = (C.c_int * (sp.world + 1))
            rc = lib.impdar_ps_alltoall_dev(ctx, d_tk.ptr, code, sp.snum, sp.tnum, sp.world, sp.rank,
                                            ia(*sp.tau_edges), ia(*sp.k_edges), d_t2.ptr)
            _hip.check(rc, 'impdar_ps_alltoall_dev')
            rc = lib.impdar_phaseshift_finish_dev(ctx, d_t2.ptr, code, tw, sp.tnum, d_out.ptr)
            _hip.check(rc, 'impdar_phaseshift_finish_dev')
            _hip.check(lib.impdar_ctx_sync(ctx), 'impdar_ctx_sync')
            return d_out.to_host()[:tw]
        finally:
            for d in (d_in, d_tk, d_t2, d_out):
                d.free()


def migrate_phaseshift_sharded(data, geometry, vconst=1.69e8, vmig=None, htaper=100, vtaper=1000, rdv=None, ctx=None,
                               engine=None):
    """Collective: every rank calls this with the WHOLE (snum, tnum) radargram and gets back
    ``(tau_lo, tau_hi, rows)``: the depth rows of the migrated image it owns, in ``data``'s dtype.

    ``geometry``: dict(snum, tnum, nt, kx, ws, dt, travel_time [us]) as migrationPhaseShift builds them
    (mig_python.py:246-263).  ``vmig``: None (constant ``vconst``) or the (snum,) profile of getVelocityProfile."""
    from . import _hip
    own_rdv = rdv is None
    rdv = Rendezvous() if rdv is None else rdv
    if engine is None:
        _hip.load()
        ctx = _hip.context() if ctx is None else ctx
        if rdv.world > 1 and not _hip.load().impdar_comm_size(ctx) == rdv.world:
            init_communicator(ctx, rdv)
    g = geometry
    sp = ShardedPhaseShift(ctx, g['snum'], g['tnum'], g['nt'], g['kx'], g['ws'], g['dt'], g['travel_time'], vconst, vmig,
                           htaper, vtaper, rdv.rank, rdv.world, np.asarray(data).dtype, engine)
    rows = sp.run(data)
    rdv.barrier()
    if own_rdv:
        rdv.close()
    return sp.tau_lo, sp.tau_hi, rows


# ---------------------------------------------------------------------------------------------------------
# single-process front door: one worker per GPU
# ---------------------------------------------------------------------------------------------------------
def ngpus_requested():
    """``$IMPDAR_NGPUS`` (0 / unset: the ordinary one-GPU path)."""
    try:
        return max(int(os.environ.get('IMPDAR_NGPUS', '0')), 0)
    except ValueError:
        return 0


def spawn_ranks(argv, world, env_extra=None, timeout=None):
    """Start ``world`` copies of ``argv`` (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE set, a job name and a
    random secret for the rendezvous) and wait for all of them.  The parent must not have touched the GPU.
    As soon as one rank exits with an error -- or ``timeout`` seconds have passed -- the others are terminated
    (a rank that died inside a collective would otherwise leave its peers waiting in RCCL for ever).
    Returns the list of return codes."""
    job = 'spawn_%d_%d' % (os.getpid(), int(time.time() * 1e3) % 1000000)
    secret = secrets.token_hex(32)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), IMPDAR_RDV_JOB=job,
                   IMPDAR_RDV_SECRET=secret)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # the parent's device choice is not the ranks': _hip.context() prefers IMPDAR_DEVICE over LOCAL_RANK, so an
        # inherited value would put every rank on one GPU (and an inherited IMPDAR_NGPUS would make the workers spawn)
        env.pop('IMPDAR_DEVICE', None)
        env.pop('IMPDAR_NGPUS', None)
        env.update(env_extra or {})
        procs.append(subprocess.Popen(argv, env=env))
    deadline = None if timeout is None else time.time() + timeout
    codes = [None] * world
    failed = False
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
                failed = failed or (codes[i] is not None and codes[i] != 0)
        if failed or (deadline is not None and time.time() > deadline):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
            t_kill = time.time() + 5.0
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(max(t_kill - time.time(), 0.1))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.02)
    return codes


def shard_timeout():
    """Seconds ``run_sharded`` waits for its workers (``$IMPDAR_SHARD_TIMEOUT``, default half an hour)."""
    try:
        return float(os.environ.get('IMPDAR_SHARD_TIMEOUT', '1800'))
    except ValueError:
        return 1800.0


def _run_workers(kind, data, arrays, meta, ngpus, timeout):
    """The job directory of ``run_sharded`` / ``run_sharded_phaseshift``: a private (mode 0700, ``mkdtemp``)
    directory of /dev/shm holding the radargram, the geometry arrays (plain ``.npy``), ``meta.json`` and the
    float64 result the workers fill; one worker process per GPU (``impdar_amd._shard_worker``)."""
    snum, tnum = data.shape
    ngpus = int(ngpus)
    if ngpus < 1:
        raise ValueError('ngpus must be at least 1, got %d' % ngpus)
    shm = '/dev/shm' if os.path.isdir('/dev/shm') else tempfile.gettempdir()
    base = tempfile.mkdtemp(prefix='impdar_shard_', dir=shm)
    f_out = os.path.join(base, 'out.npy')
    try:
        np.save(os.path.join(base, 'in.npy'), data)
        for name, a in arrays.items():
            np.save(os.path.join(base, name + '.npy'), np.asarray(a, dtype=np.float64))
        out = np.lib.format.open_memmap(f_out, mode='w+', dtype=np.float64, shape=(snum, tnum))
        del out
        with open(os.path.join(base, 'meta.json'), 'w') as fo:
            json.dump(dict(meta, kind=kind, snum=snum, tnum=tnum), fo)
        codes = spawn_ranks([sys.executable, '-m', 'impdar_amd._shard_worker', base], ngpus,
                            timeout=shard_timeout() if timeout is None else timeout)
        if any(codes):
            raise RuntimeError('sharded %s migration failed: worker exit codes %s' % (kind, codes))
        return np.array(np.load(f_out, mmap_mode='r', allow_pickle=False))
    finally:
        shutil.rmtree(base, ignore_errors=True)


def _host_radargram(data):
    data = np.ascontiguousarray(data)
    if data.dtype not in (np.float32, np.float64):
        data = data.astype(np.float64)
    return data


def run_sharded(data, dist_km, travel_time_us, vel=1.69e8, nearfield=False, ngpus=2, mode='auto', timeout=None):
    """Migrate a host radargram on ``ngpus`` GPUs from a single process: one worker process per GPU runs
    ``migrate_kirchhoff_sharded`` on its shard (``_run_workers``).  A worker that fails (among others: ``ngpus``
    larger than the number of visible GPUs) or a job that exceeds ``timeout`` seconds raises here instead of
    hanging.  Returns the float64 migrated array like migrationKirchhoff (mig_python.py:118)."""
    return _run_workers('Kirchhoff', _host_radargram(data), dict(dist=dist_km, travel_time=travel_time_us),
                        dict(vel=float(vel), nearfield=bool(nearfield), mode=str(mode)), ngpus, timeout)


def run_sharded_phaseshift(data, nt, kx, ws, dt, travel_time_us, vconst=1.69e8, vmig=None, htaper=100, vtaper=1000,
                           ngpus=2, timeout=None):
    """Phase-shift migration of a host radargram on ``ngpus`` GPUs from a single process: one worker per GPU runs
    ``migrate_phaseshift_sharded`` (wavenumber slabs -> all-to-all -> depth-row slabs).  Returns the float64
    migrated array like migrationPhaseShift (mig_python.py:282)."""
    if not np.issubdtype(np.asarray(data).dtype, np.floating):
        # as migrationPhaseShift: the reference's in-place taper cannot cast float -> int (mig_python.py:258)
        raise TypeError("Cannot cast ufunc 'multiply' output from dtype('float64') to dtype('%s') with casting rule "
                        "'same_kind'" % np.asarray(data).dtype)
    arrays = dict(kx=kx, ws=ws, travel_time=travel_time_us)
    if vmig is not None:
        arrays['vmig'] = vmig
    return _run_workers('phase-shift', _host_radargram(data), arrays,
                        dict(nt=int(nt), dt=float(dt), vconst=float(vconst), has_vmig=vmig is not None,
                             htaper=float(htaper), vtaper=float(vtaper)), ngpus, timeout)

"""Sharding of a Kirchhoff migration across the GPUs of one node.

Outputs are independent, so each rank owns a contiguous block of output
traces; the only exchange is an all-gather of the trace-major input image
(every rank prepares -- gradient + transpose -- only its own equal-width
block of input traces).  Output blocks are sized by
in-aperture pair count, not trace count: traces near the ends of the profile
see roughly half the aperture of interior traces.

Pure NumPy host logic; the collective itself is RCCL inside the C library
(``impdar_kirch_allgather``).  ``exchange_host`` is the same data movement
over ``torch.distributed`` (gloo) for CPU-side tests of the partitioning.
"""
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


def exchange_host(local_image, rank, nranks, per):
    """All-gather of per-rank (per, snum) image blocks over torch.distributed
    (any backend; used with gloo on CPU in tests).  Returns (per*nranks, snum)."""
    import torch
    import torch.distributed as dist
    block = np.zeros((per, local_image.shape[1]), dtype=local_image.dtype)
    block[:local_image.shape[0]] = local_image
    mine = torch.from_numpy(block)
    parts = [torch.empty_like(mine) for _ in range(nranks)]
    dist.all_gather(parts, mine)
    return np.concatenate([p.numpy() for p in parts], axis=0)

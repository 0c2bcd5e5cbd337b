"""Per-rank cost of the phase shift sharded over the wavenumbers, measured on ONE GPU: for N = 1, 2, 4, 8 the
stages rank r of N would run on config 5 (8192 x 8192 float32, v(z) of 4 layers and constant velocity):
    tk      impdar_phaseshift_tk_dev   replicated transforms + frequency sums of tnum/N wavenumbers
    pack    the pack kernels of impdar_ps_alltoall_dev (world 1 call on a [tnum/N][snum] slab: pack + device copy)
    finish  impdar_phaseshift_finish_dev on snum/N depth rows
and the bytes the rank would put on xGMI.  `projected_ms` = tk + pack + finish + bytes / (7 links x 40 GB/s), a
conservative ring-free point-to-point figure; it is a projection, not a measurement of the exchange.
usage: python3 profiles/tools/ps_sharded_emulate.py [n=8192]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from impdar_amd import _hip as hip, parallel, synth      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
lib = hip.load()
ctx = hip.context()
snum = tnum = n
geo = synth.geometry(snum, tnum, dx=2.0)
data = synth.noise_radargram(snum, tnum, seed=1).astype(np.float32)
nt = n
dist = geo['dist']
kx = 2. * np.pi * np.fft.fftfreq(tnum, d=2.0)
ws = 2. * np.pi * np.fft.fftfreq(nt, d=geo['dt'])
layers = np.repeat([1.69e8, 1.8e8, 1.95e8, 2.1e8], snum // 4).astype(np.float64)
d_in = hip.DeviceArray.from_host(ctx, data)
p = lambda a: hip.as_dp(a)[1]


def timed(fn, reps=3):
    fn()
    hip.check(lib.impdar_ctx_sync(ctx))
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        hip.check(lib.impdar_ctx_sync(ctx))
        t.append((time.perf_counter() - t0) * 1e3)
    return min(t)


for name, vm in (('vz', layers), ('const', None)):
    d_full = hip.DeviceArray(ctx, (snum, tnum), np.float32)
    base = timed(lambda: hip.check(lib.impdar_phaseshift_dev(
        ctx, d_in.ptr, hip.F32, snum, tnum, nt, p(kx), p(ws), geo['dt'], p(geo['travel_time']),
        1.69e8 if vm is None else 0.0, None if vm is None else p(vm), 0 if vm is None else snum, 100.0, 1000.0, d_full.ptr)))
    d_full.free()
    for world in (1, 2, 4, 8):
        ke, te = parallel.slab_edges(tnum, world), parallel.slab_edges(snum, world)
        r = world // 2                       # a middle rank (its slab holds the high wavenumbers for world > 1)
        nk, tw = ke[r + 1] - ke[r], te[r + 1] - te[r]
        d_tk = hip.DeviceArray(ctx, (nk, snum), np.complex64)
        d_cp = hip.DeviceArray(ctx, (nk, snum), np.complex64)
        d_t2 = hip.DeviceArray(ctx, (tnum, tw), np.complex64)
        d_out = hip.DeviceArray(ctx, (tw, tnum), np.float32)
        t_tk = {}
        for rr in sorted({0, r}):
            t_tk[rr] = timed(lambda: hip.check(lib.impdar_phaseshift_tk_dev(
                ctx, d_in.ptr, hip.F32, snum, tnum, nt, p(kx), p(ws), geo['dt'], p(geo['travel_time']),
                1.69e8 if vm is None else 0.0, None if vm is None else p(vm), 0 if vm is None else snum, 100.0, 1000.0,
                ke[rr], ke[rr + 1] - ke[rr], d_tk.ptr)))
        # the pack stage: a one-rank exchange of an [nk][snum] slab (pack kernel + device copy of nk*snum*8 bytes)
        ia = C.c_int * 2
        t_pack = timed(lambda: hip.check(lib.impdar_ps_alltoall_dev(ctx, d_tk.ptr, hip.F32, snum, nk, 1, 0, ia(0, snum),
                                                                    ia(0, nk), d_cp.ptr)))
        t_fin = timed(lambda: hip.check(lib.impdar_phaseshift_finish_dev(ctx, d_t2.ptr, hip.F32, tw, tnum, d_out.ptr)))
        sent = nk * (snum - tw) * 8
        wire = sent / (7 * 40e9) * 1e3 if world > 1 else 0.0
        worst = max(t_tk.values())
        proj = worst + t_pack + t_fin + wire
        print(json.dumps(dict(case=name, n=n, world=world, unsharded_ms=round(base, 2),
                              tk_ms={str(k): round(v, 2) for k, v in t_tk.items()}, pack_ms=round(t_pack, 2),
                              finish_ms=round(t_fin, 2), sent_MB=round(sent / 1e6, 1), wire_ms=round(wire, 2),
                              projected_ms=round(proj, 2), projected_speedup=round(base / proj, 2))), flush=True)
        for d in (d_tk, d_cp, d_t2, d_out):
            d.free()

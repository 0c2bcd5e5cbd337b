#!/usr/bin/env python3
"""One-off randomized sweep of the wavenumber-sharded phase shift on ONE GPU (not part of the pytest suites): random
sizes, rank counts 1..8 (also more ranks than wavenumbers), float32 / float64, constant velocity and layered profiles.
For every case the slabs impdar_phaseshift_tk_dev returns must be BIT-EQUAL to the same rows of the unsharded TK, and
the image assembled from impdar_phaseshift_finish_dev of every rank's depth rows (all-to-all carried out on the host
from parallel.alltoall_layout) must equal the unsharded call within rocFFT's rounding (2e-6 float32, 1e-13 float64).

    python tests/tools/fuzz_ps_sharded.py [ncases] [seed]  ->  one line per case, summary, exit code 1 on a miss
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from impdar_amd import _hip as hip, parallel, synth              # noqa: E402
from oracle import mig_oracle                                    # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    lib, ctx = hip.load(), hip.context()
    bad, worst = 0, {np.float32: 0.0, np.float64: 0.0}
    t_start = time.time()
    p = lambda a: hip.as_dp(a)[1]
    for case in range(ncases):
        snum = int(rng.choice([40, 97, 128, 256, 300, 333, 512, 700, 1024]))
        tnum = int(rng.choice([1, 2, 5, 16, 33, 64, 100, 256, 512, 600]))
        world = int(rng.integers(1, 9))
        dtype = np.float32 if rng.random() < 0.6 else np.float64
        kind = int(rng.integers(0, 3))
        geo = synth.geometry(snum, tnum, dx=float(rng.choice([1.0, 2.0, 4.0])))
        data = synth.noise_radargram(snum, tnum, seed=int(rng.integers(1 << 30))).astype(dtype)
        nt = int(2 ** np.ceil(np.log(snum) / np.log(2)))
        kx = mig_oracle._kx(tnum, geo['trace_int'], geo['dist'])
        ws = 2. * np.pi * np.fft.fftfreq(nt, d=geo['dt'])
        vm = None
        if kind == 1:          # a few long runs (the matrix-core kernel for float32 when the size allows)
            cuts = np.sort(rng.choice(np.arange(1, snum), size=int(rng.integers(1, 4)), replace=False))
            vm = np.full(snum, 1.69e8)
            for i, c in enumerate(cuts):
                vm[c:] = 1.69e8 * (1.0 + 0.07 * (i + 1))
        elif kind == 2:        # a velocity that changes at every step (per-step kernels)
            vm = np.linspace(1.6e8, 2.2e8, snum)
        vconst, vlen = (1.69e8, 0) if vm is None else (0.0, snum)
        code = hip.dtype_code(dtype)
        cdt = np.complex64 if dtype == np.float32 else np.complex128
        d_in = hip.DeviceArray.from_host(ctx, data)
        d_out = hip.DeviceArray(ctx, (snum, tnum), dtype)
        hip.check(lib.impdar_phaseshift_dev(ctx, d_in.ptr, code, snum, tnum, nt, p(kx), p(ws), geo['dt'], p(geo['travel_time']),
                                            vconst, None if vm is None else p(vm), vlen, 20.0, 30.0, d_out.ptr), 'unsharded')
        want = d_out.to_host().astype(np.float64)
        d_out.free()

        def tk(k0, nk):
            d_tk = hip.DeviceArray(ctx, (max(nk, 1), snum), cdt)
            hip.check(lib.impdar_phaseshift_tk_dev(ctx, d_in.ptr, code, snum, tnum, nt, p(kx), p(ws), geo['dt'],
                                                   p(geo['travel_time']), vconst, None if vm is None else p(vm), vlen, 20.0, 30.0,
                                                   k0, nk, d_tk.ptr), 'tk')
            a = d_tk.to_host()[:nk]
            d_tk.free()
            return a
        full = tk(0, tnum)
        ke, te = parallel.slab_edges(tnum, world), parallel.slab_edges(snum, world)
        esz = full.itemsize
        ok_slabs, sbufs = True, []
        for r in range(world):
            slab = tk(ke[r], ke[r + 1] - ke[r])
            ok_slabs = ok_slabs and np.array_equal(slab.view(np.uint8), full[ke[r]:ke[r + 1]].view(np.uint8))
            sbufs.append(np.concatenate([slab[:, te[s]:te[s + 1]].ravel() for s in range(world)] + [np.zeros(0, cdt)]).view(np.uint8))
        got = np.zeros((snum, tnum))
        for r in range(world):
            tw = te[r + 1] - te[r]
            if tw == 0:
                continue
            rbuf = np.full(tnum * tw * esz, 0xA5, dtype=np.uint8)
            _, recv = parallel.alltoall_layout(te, ke, r, esz)
            for s, roff, rn in recv:
                _, soff, sn = parallel.alltoall_layout(te, ke, s, esz)[0][r]
                rbuf[roff:roff + rn] = sbufs[s][soff:soff + sn]
            d_t2 = hip.DeviceArray.from_host(ctx, rbuf.view(cdt).reshape(tnum, tw))
            d_rows = hip.DeviceArray(ctx, (tw, tnum), dtype)
            hip.check(lib.impdar_phaseshift_finish_dev(ctx, d_t2.ptr, code, tw, tnum, d_rows.ptr), 'finish')
            got[te[r]:te[r + 1]] = d_rows.to_host()
            d_t2.free()
            d_rows.free()
        d_in.free()
        scale = max(np.max(np.abs(want)), 1e-300)
        err = float(np.max(np.abs(got - want)) / scale)
        tol = 2e-6 if dtype == np.float32 else 1e-13
        ok = ok_slabs and err < tol
        bad += 0 if ok else 1
        worst[dtype] = max(worst[dtype], err)
        print('%3d %s snum %4d tnum %3d world %d kind %d slabs %s err %.2e %s'
              % (case, np.dtype(dtype).name, snum, tnum, world, kind, 'bit-equal' if ok_slabs else 'DIFFER', err, 'ok' if ok else 'MISS'), flush=True)
    print('cases %d, misses %d, worst float32 rel-max %.2e (bar 2e-6), worst float64 rel-max %.2e (bar 1e-13), %.0f s'
          % (ncases, bad, worst[np.float32], worst[np.float64], time.time() - t_start))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()

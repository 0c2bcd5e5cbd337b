"""NumPy restatement of csrc/ps_series.h (test infrastructure): the phase shift's frequency sum over a velocity that changes at
every step, piece by piece -- frequencies near the evanescent boundary summed directly, the others through J non-uniform DFTs
that share their nodes, with the pieces, cuts and series tables of the library's OWN planner (csrc/ps_series_plan.h, compiled by
itself with g++ -- no GPU, no HIP)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MSER, NKX = 16, 16
W32, W64 = 8, 14


def planner(tmpdir):
    src = os.path.join(tmpdir, 'sr_probe.cpp')
    lib = os.path.join(tmpdir, 'libsrplan.so')
    with open(src, 'w') as f:
        f.write('#define SR_PLAN_PROBE 1\n#include "ps_series_plan.h"\n')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'impdar_amd', 'csrc'), src, '-o', lib])
    return C.CDLL(lib)


def plan(lib, v, dt, dw, nf, kxh_max, dbl):
    v = np.ascontiguousarray(v, dtype=np.float64)
    cap, evcap = 4096, len(v) * 8 + 64
    ints = np.zeros((cap, 6), dtype=np.int32)
    dbls = np.zeros((cap, 2 + MSER + NKX))
    ev = np.zeros(evcap)
    offs = np.zeros(cap, dtype=np.int32)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    n = lib.impdar_sr_plan_probe(v.ctypes.data_as(dp), len(v), C.c_double(dt), C.c_double(dw), nf, C.c_double(kxh_max), int(dbl), cap,
                                 ints.ctypes.data_as(ip), dbls.ctypes.data_as(dp), ev.ctypes.data_as(dp), evcap, offs.ctypes.data_as(ip))
    assert n > 0, n
    plan.model_cost = float(ev[int(offs[n - 1]) + int(ints[n - 1, 1]) * int(ints[n - 1, 5])])
    out = []
    for i in range(n):
        start, ln, loglp, J, mser, mj = (int(x) for x in ints[i])
        out.append(dict(start=start, len=ln, loglp=loglp, J=J, mser=mser, mj=mj, vb2=dbls[i, 0], s=dbls[i, 1], be=dbls[i, 2:2 + MSER].copy(),
                        lam=dbls[i, 2 + MSER:].copy(), ev=ev[offs[i]:offs[i] + ln * mj].reshape(ln, mj) if mj else np.zeros((ln, 0))))
    return out


def b_coef(m):
    b = 0.5
    for q in range(1, m):
        b *= (2 * q - 1) / (2 * q + 2)
    return b


def direct_sum(F, w, c2, dt):
    """mig_python.py:438-487 for one wavenumber: TK[tau] = sum_w alive FK exp(i Phi_tau).  The phases are summed in extended precision
    (a float64 running sum of 8192 phases of up to pi carries ~1e-9 of rounding noise -- the reference itself multiplies unit
    complex numbers step by step and stays at 1e-14)"""
    arg = w[None, :] ** 2 - c2[:, None]
    alive = np.logical_and.accumulate(arg > 0, axis=0)
    ph = np.cumsum(np.longdouble(dt) * np.sqrt(np.maximum(arg, 0).astype(np.longdouble)), axis=0)
    ph = (ph - 2 * np.pi * np.rint(ph / (2 * np.longdouble(np.pi)))).astype(np.float64) if False else np.remainder(ph, 2 * np.longdouble(3.14159265358979323846264338327950288)).astype(np.float64)
    return np.where(alive, F[None, :] * np.exp(1j * ph), 0).sum(1)


def wrap(x):
    return x - 6.283185307179586 * np.rint(x * 0.15915494309189535)


def window(x, w, dtype):
    z = np.maximum(1 - (2 * x / w) ** 2, 0).astype(dtype)
    return np.exp((2.30 * w * (np.sqrt(z) - 1)).astype(dtype)).astype(dtype)


def correction(lp, w):
    g, ns = 2 * lp, 512
    x = np.linspace(-w / 2, w / 2, ns + 1)
    wq = np.ones(ns + 1)
    wq[1:-1:2], wq[2:-1:2] = 4, 2
    psi = np.exp(2.30 * w * (np.sqrt(np.maximum(1 - (2 * x / w) ** 2, 0)) - 1)) * wq
    n = np.arange(-lp // 2, lp // 2)
    return 1.0 / ((psi[None, :] * np.cos(2 * np.pi * np.outer(n, x) / g)).sum(1) * (w / ns) / 3)


def series_sum(pieces, F, w, v, kx, dt, kxh_max, dtype=None, stats=None):
    """The scheme on one wavenumber.  dtype None: the J transforms evaluated exactly (the series alone); float32 / float64: through
    the window + FFT in that arithmetic (W = 8 / 14)."""
    n = len(v)
    kxh = 0.5 * abs(kx)
    c2 = (0.5 * v * kx) ** 2                                       # (life and death: the oracle's own rounding)
    out = np.zeros(n, dtype=complex)
    ph = np.zeros(len(w))
    alive = np.ones(len(w), dtype=bool)
    jk = min(max(int(np.ceil(kxh / kxh_max * NKX)) - 1, 0), NKX - 1)
    nd = nr = 0
    for pc in pieces:
        a, L, J, mj = pc['start'], pc['len'], pc['J'], pc['mj']
        lam = pc['lam'][jk]
        psi_min = kxh * lam
        cb2 = kxh * kxh * pc['vb2']
        psi2 = w * w - cb2
        reg = alive & (psi2 > psi_min ** 2) & (psi2 > 4e-8 * w * w)
        dirs = alive & ~reg
        seg = np.zeros(L, dtype=complex)
        if dirs.any():
            arg = w[None, dirs] ** 2 - c2[a:a + L, None]
            al = np.logical_and.accumulate(arg > 0, axis=0)
            phd = (ph[None, dirs] + np.cumsum(np.longdouble(dt) * np.sqrt(np.maximum(arg, 0).astype(np.longdouble)), axis=0))
            phd = np.remainder(phd, 2 * np.longdouble(3.14159265358979323846264338327950288)).astype(np.float64)
            seg += np.where(al, F[None, dirs] * np.exp(1j * phd), 0).sum(1)
            ph[dirs] = wrap(phd[-1])
            alive[dirs] = al[-1]
        nd += int(dirs.sum()) * L
        nr += int(reg.sum())
        if reg.any():
            psi = np.sqrt(psi2[reg])
            inc = dt * psi
            z = psi_min / psi
            rho = pc['s'] / (lam * lam) if lam > 0 else 0.0
            y = np.zeros((J, L), dtype=complex)
            y[0] = 1
            if J > 1:
                cm = np.array([-dt * psi_min * b_coef(m) * rho ** m for m in range(1, mj + 1)])
                r = cm[None, :] * pc['ev']                           # (L, mj): coefficient of z^(2m-1)
                for p in range(1, J):
                    acc = 0
                    for m in range(mj):
                        k = 2 * m + 1
                        if k <= p:
                            acc = acc + k * 1j * r[:, m] * y[p - k]
                    y[p] = acc / p
            Cw = F[reg] * np.exp(1j * ph[reg])
            if dtype is None:
                Eb = np.exp(1j * np.outer(np.arange(1, L + 1), inc))
                Cz = Cw.copy()
                for p in range(J):
                    seg += y[p] * (Eb @ Cz)
                    Cz = Cz * z
            else:
                cdt = np.complex64 if dtype == np.float32 else np.complex128
                ww = W32 if dtype == np.float32 else W64
                lp = 1 << pc['loglp']
                g = 2 * lp
                d = (Cw * np.exp(1j * inc * (1 + lp / 2))).astype(cdt)
                u = inc * g / (2 * np.pi)
                m0 = np.floor(u).astype(int)
                fr = (u - m0).astype(dtype)
                zz = z.astype(dtype)
                corr = correction(lp, ww)
                npr = np.arange(L) - lp // 2
                dz = d.copy()
                for p in range(J):
                    grid = np.zeros(g, dtype=cdt)
                    for dm in range(-ww // 2 + 1, ww // 2 + 1):
                        np.add.at(grid, (m0 + dm) % g, (dz * window(fr - dtype(dm), ww, dtype)).astype(cdt))
                    ghat = (np.fft.ifft(grid) * g).astype(cdt)
                    seg += y[p].astype(cdt) * (ghat[npr % g] * corr[npr + lp // 2].astype(dtype))
                    dz = (dz * zz).astype(cdt)
            t = kxh * kxh * pc['s'] / psi2[reg]
            ser = np.zeros_like(t)
            for m in range(pc['mser'] - 1, -1, -1):
                ser = ser * t + pc['be'][m]
            ser = ser * t
            ph[reg] = wrap(ph[reg] + inc * (L - ser))
        out[a:a + L] = seg
    if stats is not None:
        stats['direct_pairs'] = nd
        stats['regular'] = nr
    return out

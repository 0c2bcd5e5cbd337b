"""Round 6, prototype (NumPy, CPU): the phase shift's frequency sum for a velocity that CHANGES inside a piece, as a few
non-uniform FFTs of one piece.

Reference loop (mig_python.py:438-487): TK[tau] = sum_w alive * FK_w * exp(i Phi_tau(w)),  Phi_tau = sum_{t <= tau} phi_t(w),
phi_t(w) = dt sqrt(w^2 - c_t^2), c_t = v_t kx / 2 (sticky zero once w^2 <= c_t^2).

Inside a piece [a, a + L) write c_t^2 = cbar^2 + eps_t, psi_w = sqrt(w^2 - cbar^2):
    phi_t = dt psi sqrt(1 - eps_t / psi^2) = dt psi - dt sum_m b_m eps_t^m psi^(1-2m),    b = 1/2, 1/8, 1/16, 5/128, ...
    Phi(a + n) = Phi(a - 1) + (n + 1) dt psi + R(n, w),   R = -dt sum_m b_m E_m(n) psi^(1-2m),  E_m(n) = sum_{t=a..a+n} eps_t^m
    exp(i R) = sum_p y_p(n) z^p,  z = psi_min / psi <= 1     (power series of exp of a polynomial in z: a recurrence per n)
so  TK[a + n] = sum_{p < J} y_p(n) * [ sum_w (C_w z_w^p) e^{i (n+1) dt psi_w} ]  -- J type-1 non-uniform DFTs that share their
nodes (and so their window values); frequencies with psi < psi_min (near the evanescent boundary, where 1/psi blows up) are
summed directly.  psi_min is chosen per (piece, kx) so that a majorant of the series' tail at z = 1 stays under `tol`.

This script measures, on 8192-step profiles, the error of the truncated series against the float64 direct sum as a function of
(L, J, tol), the share of frequencies that go to the direct sum, and a cost model.  `--nufft` also runs the window/FFT
evaluation in float32 to show the combined error.
"""
import argparse
import sys
import numpy as np

B = [0.0]
for m in range(1, 40):          # sqrt(1 - x) = 1 - sum b_m x^m
    B.append((B[-1] * (2 * m - 3) / (2 * m)) if m > 1 else 0.5)


def profiles(n, dt):
    u = np.linspace(0., 1., n)
    tt = np.arange(n) * dt
    out = {
        'gradient': 1.69e8 + 0.5e8 * u,
        'wavy': 1.8e8 + 0.15e8 * np.sin(7. * u) + 0.1e8 * u,
        'firn': 1.69e8 + 0.6e8 * np.exp(-tt / 0.8e-6),                      # fast near the surface, flat below
        'stairs41': 1.69e8 + 0.51e8 * np.floor(u * 40) / 40,
        'noisy_const': 1.69e8 * (1 + 4e-13 * np.random.default_rng(1).standard_normal(n)),
    }
    return out


def direct(F, w, c2, dt):
    """TK[tau] and the per-frequency terms' alive mask; float64.  F, w: (nf,), c2: (n,)"""
    n = len(c2)
    arg = w[None, :] ** 2 - c2[:, None]
    alive = np.logical_and.accumulate(arg > 0, axis=0)
    ph = np.cumsum(dt * np.sqrt(np.maximum(arg, 0)), axis=0)
    return (np.where(alive, F[None, :] * np.exp(1j * ph), 0)).sum(1), alive, ph


def majorant_tail(rhat, J, P):
    """sum_{p >= J} of the coefficients of exp(sum_m rhat_m z^(2m-1)) at z = 1 (a majorant of the series of exp(i R))"""
    q = np.zeros(P)
    for m, r in enumerate(rhat, start=1):
        if 2 * m - 1 < P:
            q[2 * m - 1] = r
    y = np.zeros(P)
    y[0] = 1
    for p in range(1, P):
        y[p] = sum(k * q[k] * y[p - k] for k in range(1, p + 1, 2)) / p
    return y[J:].sum()


def cut_lambda(dt, kxh, Ev_abs, emax_v, J, tol, rho_max, mterms):
    """smallest lam (psi_min = kxh * lam, kxh = kx / 2) with majorant tail <= tol and eps / psi_min^2 <= rho_max: bisection in log(lam)"""
    if kxh == 0 or Ev_abs[0] == 0:
        return np.sqrt(emax_v / rho_max) if emax_v > 0 else 0.0
    lo = np.sqrt(emax_v / rho_max)
    def tail(lam):
        rhat = [dt * kxh * B[m] * Ev_abs[m - 1] * lam ** (1 - 2 * m) for m in range(1, mterms + 1)]
        return majorant_tail(rhat, J, J + 2 * mterms)
    if tail(lo) <= tol:
        return lo
    hi = lo
    while tail(hi) > tol:
        hi *= 2
    for _ in range(30):
        mid = np.sqrt(lo * hi)
        if tail(mid) > tol:
            lo = mid
        else:
            hi = mid
    return hi


def series_piece(F, w, v2p, kxh, dt, ph0, alive0, J, tol, rho_max=0.1, mterms=12, c2_exact=None):
    """One piece: returns (TK piece, phase at the end, alive at the end, n_direct, n_regular, tail estimate)."""
    L = len(v2p)
    vb2 = v2p.mean()
    ev = v2p - vb2                                                              # in velocity^2; eps = kxh^2 ev
    Ev = np.stack([np.cumsum(ev ** m) for m in range(1, mterms + 1)])            # (M, L)
    lam = cut_lambda(dt, kxh, np.abs(Ev).max(1), np.abs(ev).max(), J, tol, rho_max, mterms)
    psi_min = max(kxh * lam, 1e-3)
    c2p = kxh * kxh * v2p if c2_exact is None else c2_exact      # (the direct set decides life and death: the oracle's own rounding)
    cb2 = kxh * kxh * vb2
    E = np.stack([kxh ** (2 * m) * Ev[m - 1] for m in range(1, mterms + 1)])
    psi2 = w * w - cb2
    reg = alive0 & (psi2 > psi_min ** 2)
    dirs = alive0 & ~reg
    out = np.zeros(L, dtype=complex)
    ph_end = ph0.copy()
    alive_end = alive0.copy()
    if dirs.any():
        arg = w[None, dirs] ** 2 - c2p[:, None]
        al = np.logical_and.accumulate(arg > 0, axis=0)
        ph = ph0[None, dirs] + np.cumsum(dt * np.sqrt(np.maximum(arg, 0)), axis=0)
        out += np.where(al, F[None, dirs] * np.exp(1j * ph), 0).sum(1)
        ph_end[dirs] = ph[-1]
        alive_end[dirs] = al[-1]
    tail = 0.0
    if reg.any():
        psi = np.sqrt(psi2[reg])
        z = psi_min / psi
        P = J
        q = np.zeros((P, L), dtype=complex)
        for m in range(1, mterms + 1):
            k = 2 * m - 1
            if k < P and np.abs(E[m - 1]).max() > 0:
                q[k] = 1j * (-dt * B[m] * E[m - 1] * psi_min ** (1 - 2 * m))
        y = np.zeros((P, L), dtype=complex)
        y[0] = 1
        for p in range(1, P):
            s = 0
            for k in range(1, p + 1, 2):
                s = s + k * q[k] * y[p - k]
            y[p] = s / p
        C = F[reg] * np.exp(1j * ph0[reg])
        nn = np.arange(1, L + 1)
        Eb = np.exp(1j * np.outer(nn, dt * psi))                                   # (L, nreg): the J transforms, evaluated exactly
        Cz = C.copy()
        for p in range(J):
            out += y[p] * (Eb @ Cz)
            Cz = Cz * z
        r_end = -dt * sum(B[m] * E[m - 1][-1] * psi ** (1 - 2 * m) for m in range(1, mterms + 1))
        ph_end[reg] = ph0[reg] + L * dt * psi + r_end
    return out, ph_end, alive_end, int(dirs.sum()), int(reg.sum()), lam


def run(name, v, ks, n, dt, L, J, tol, rng):
    nt = n
    wall = 2 * np.pi * np.fft.fftfreq(nt, d=dt)
    w = np.abs(wall[1:nt // 2 + 1])
    kx = 2 * np.pi * np.fft.fftfreq(n, d=1.0)
    rows = []
    for k in ks:
        c2 = (0.5 * v * kx[k]) ** 2
        F = rng.standard_normal(len(w)) + 1j * rng.standard_normal(len(w))
        want, alive, _ = direct(F, w, c2, dt)
        got = np.zeros(n, dtype=complex)
        ph = np.zeros(len(w))
        al = np.ones(len(w), dtype=bool)
        nd = nr = 0
        tail = 0.
        for a in range(0, n, L):
            o, ph, al, d_, r_, t_ = series_piece(F, w, (v * v)[a:a + L], 0.5 * kx[k], dt, ph, al, J, tol, c2_exact=c2[a:a + L])
            got[a:a + L] = o
            nd += d_ * min(L, n - a)
            nr += r_
            tail = max(tail, t_)
        err = np.linalg.norm(got - want) / np.linalg.norm(want)
        emax = np.abs(got - want).max() / np.abs(want).max()
        rows.append((k, err, emax, tail, nd, nr))
    return rows


def cost(nd, nr, L, J, Wn, direct_instr, win_instr):
    """vector instructions per wavenumber: direct (step, frequency) pairs + per regular (frequency, piece): coefficient ~80,
    window values Wn * (win_instr + 4 J)"""
    return nd * direct_instr + nr * (80 + Wn * (win_instr + 4 * J))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=8192)
    ap.add_argument('--profiles', default='gradient,wavy,firn,stairs41')
    ap.add_argument('--ks', default='37,700,2000,3300')
    ap.add_argument('--L', default='64,128,256,512')
    ap.add_argument('--J', default='4,6,8,12,16')
    ap.add_argument('--tol', default='1e-5')
    ap.add_argument('--direct-instr', type=float, default=60.)
    a = ap.parse_args()
    n, dt = a.n, 1e-8
    prof = profiles(n, dt)
    ks = [int(x) for x in a.ks.split(',')]
    print('# n = %d, dt = %g; error = rel L2 of the piece-wise series against the float64 direct sum, worst of wavenumbers %s' % (n, dt, ks))
    print('# tol = majorant bound of the series tail at the cut (relative error of the WORST regular frequency)')
    print('# direct%% = share of live (step, frequency) pairs summed directly; cost = model instructions (direct pairs at %g) / '
          'everything direct at 12.5 per pair (ps_smooth32_kernel)' % a.direct_instr)
    print('%-10s %5s %3s %7s  %9s %9s  %7s %6s' % ('profile', 'L', 'J', 'tol', 'relL2', 'relmax', 'direct%', 'cost'))
    for name in a.profiles.split(','):
        v = prof[name]
        for L in [int(x) for x in a.L.split(',')]:
            for tol in [float(x) for x in a.tol.split(',')]:
                for J in [int(x) for x in a.J.split(',')]:
                    rows = run(name, v, ks, n, dt, L, J, tol, np.random.default_rng(5))
                    err = max(r[1] for r in rows)
                    emx = max(r[2] for r in rows)
                    nd = sum(r[4] for r in rows)
                    nr = sum(r[5] for r in rows)
                    alive_pairs = nd + nr * L
                    c = cost(nd, nr, L, J, 8, a.direct_instr, 12) / max(alive_pairs * 12.5, 1)
                    print('%-10s %5d %3d %7.0e  %9.2e %9.2e  %7.2f %6.3f' % (name, L, J, tol, err, emx, 100. * nd / max(alive_pairs, 1), c), flush=True)

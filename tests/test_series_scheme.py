"""The scheme of csrc/ps_series.h on the CPU: the phase shift's frequency sum (mig_python.py:438-487) over a velocity that changes
at every step, piece by piece -- J non-uniform DFTs that share their nodes for the regular frequencies, direct sums in the band
above the evanescent boundary -- with the pieces, cuts and series tables of the library's own planner (csrc/ps_series_plan.h,
compiled here by itself), against a float64 direct sum.  Pins: the series and its recurrence, the cut rule (majorant of the tail
<= 1e-5 / 1e-11), the phase series at the end of a piece, W = 8 / float32 and W = 14 / float64 windows on these nodes.  The kernel
itself is held to the oracle by the GPU tests."""
import numpy as np
import pytest

import series_scheme as S


@pytest.fixture(scope='module')
def lib(tmp_path_factory):
    return S.planner(str(tmp_path_factory.mktemp('srplan')))


def _axes(n, dt):
    ws = 2 * np.pi * np.fft.fftfreq(n, d=dt)
    w = np.abs(ws[1:n // 2])
    kx = 2 * np.pi * np.fft.fftfreq(n, d=1.0)
    return w, kx


PROFILES = {
    'gradient': lambda u: 1.69e8 + 0.5e8 * u,
    'falling': lambda u: 2.2e8 - 0.5e8 * u,
    'wavy': lambda u: 1.8e8 + 0.15e8 * np.sin(7. * u) + 0.1e8 * u,
    'firn': lambda u: 1.69e8 + 0.6e8 * np.exp(-u / 0.02),
    'noisy_run': lambda u: 1.69e8 * (1 + 4e-13 * np.random.default_rng(1).standard_normal(len(u))),
}


@pytest.mark.parametrize('dbl', [False, True])
@pytest.mark.parametrize('name', sorted(PROFILES))
def test_series_of_a_changing_velocity_against_the_direct_sum(lib, name, dbl):
    n, dt = 2048, 1e-8
    w, kx = _axes(n, dt)
    v = np.ascontiguousarray(PROFILES[name](np.linspace(0., 1., n)))
    kxh_max = 0.5 * np.abs(kx).max()
    pieces = S.plan(lib, v, dt, w[0], n // 2, kxh_max, dbl)
    assert sum(p['len'] for p in pieces) == n and pieces[0]['start'] == 0
    rng = np.random.default_rng(7)
    worst = 0.0
    for k in (9, 170, 500, 830):
        F = rng.standard_normal(len(w)) + 1j * rng.standard_normal(len(w))
        want = S.direct_sum(F, w, (0.5 * v * kx[k]) ** 2, dt)
        got = S.series_sum(pieces, F, w, v, kx[k], dt, kxh_max, None)
        worst = max(worst, np.abs(got - want).max() / np.abs(want).max())
    # the tail bound holds the WORST frequency to 1e-5 / 1e-11: the sums come out far inside
    assert worst < (5e-12 if dbl else 2e-6), worst


@pytest.mark.parametrize('dbl', [False, True])
def test_series_through_the_window_and_fft(lib, dbl):
    n, dt = 1024, 1e-8
    w, kx = _axes(n, dt)
    v = np.ascontiguousarray(PROFILES['wavy'](np.linspace(0., 1., n)))
    kxh_max = 0.5 * np.abs(kx).max()
    pieces = S.plan(lib, v, dt, w[0], n // 2, kxh_max, dbl)
    rng = np.random.default_rng(11)
    for k in (21, 300):
        F = rng.standard_normal(len(w)) + 1j * rng.standard_normal(len(w))
        want = S.direct_sum(F, w, (0.5 * v * kx[k]) ** 2, dt)
        got = S.series_sum(pieces, F, w, v, kx[k], dt, kxh_max, np.float64 if dbl else np.float32)
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err < (5e-12 if dbl else 3e-6), err


def test_a_pair_of_wavenumbers_shares_the_series_because_its_coefficients_alternate_between_real_and_imaginary():
    """ps_series_kernel<float, true> (round 6): rows k and tnum - k as G = (TK[k] + conj TK[tnum - k]) / 2.  The series' coefficients
    y_p(n) of exp(i sum_m r_m z^(2m-1)) are real for even p and imaginary for odd p (the exponent is imaginary and odd in z), so
    conj(sum_p y_p z^p S_p) = sum_p y_p (-z)^p conj(S_p): the partner's coefficients enter grid p mirrored (node -phi) with (-z)^p and the
    SAME y_p multiply the J transforms of the pair."""
    rng = np.random.default_rng(11)
    L, J, mj, nw = 40, 12, 6, 25
    r = 0.3 * rng.standard_normal((L, mj))                        # (L, mj): coefficient of z^(2m-1) at step n, real
    y = np.zeros((J, L), dtype=complex)
    y[0] = 1
    for p in range(1, J):
        acc = 0
        for m in range(mj):
            k = 2 * m + 1
            if k <= p:
                acc = acc + k * 1j * r[:, m] * y[p - k]
        y[p] = acc / p
    assert np.abs(y[0::2].imag).max() == 0.0 and np.abs(y[1::2].real).max() == 0.0
    phi = rng.uniform(0.1, 3.0, nw)
    z = rng.uniform(0.2, 1.0, nw)
    a = rng.standard_normal(nw) + 1j * rng.standard_normal(nw)     # row k
    b = rng.standard_normal(nw) + 1j * rng.standard_normal(nw)     # row tnum - k
    n1 = np.arange(1, L + 1)
    Ep = np.exp(1j * np.outer(n1, phi))
    row_k = sum(y[p] * (Ep @ (a * z ** p)) for p in range(J))
    row_m = sum(y[p] * (Ep @ (b * z ** p)) for p in range(J))
    want = 0.5 * (row_k + np.conj(row_m))
    got = 0.5 * sum(y[p] * (Ep @ (a * z ** p) + np.conj(Ep) @ (np.conj(b) * (-z) ** p)) for p in range(J))
    assert np.max(np.abs(got - want)) < 1e-12 * np.max(np.abs(want))

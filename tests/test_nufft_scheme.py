"""The scheme of csrc/ps_nufft.h restated in NumPy (CPU): inside a run of constant velocity the phase shift's frequency sum
TK[tau0 + n] = sum_w C_w e^{i phi_w (n + 1)} (mig_python.py:418-420, :464, :487) is a type-1 non-uniform DFT; an 8-point
"exponential of semicircle" window, twofold oversampling and float32 arithmetic reproduce the float64 direct sum to ~4e-7.
This pins the parameters the kernel is built on (window width, beta, oversampling, the centred band, the correction table);
the kernel itself is held to the oracle by the GPU tests."""
import numpy as np
import pytest

SIGMA = 2


def window(x, dtype, W=8):
    z = (1 - (2 * x / W) ** 2).astype(dtype)
    return np.where(z > 0, np.exp((2.30 * W * (np.sqrt(np.maximum(z, 0)) - 1)).astype(dtype)), 0).astype(dtype)


def correction(lp, W=8):
    """1 / psihat(n), n = -Lp/2 .. Lp/2 - 1: Simpson in float64, as the library's host side"""
    g, ns = SIGMA * lp, 512
    x = np.linspace(-W / 2, W / 2, ns + 1)
    wq = np.ones(ns + 1)
    wq[1:-1:2], wq[2:-1:2] = 4, 2
    psi = np.exp(2.30 * W * (np.sqrt(np.maximum(1 - (2 * x / W) ** 2, 0)) - 1)) * wq
    n = np.arange(-lp // 2, lp // 2)
    return 1.0 / ((psi[None, :] * np.cos(2 * np.pi * np.outer(n, x) / g)).sum(1) * (W / ns) / 3)


@pytest.mark.parametrize('W,dtype,cdtype,bar', [(8, np.float32, np.complex64, 1.5e-6), (14, np.float64, np.complex128, 5e-12)])
@pytest.mark.parametrize('k', [0, 37, 700])
@pytest.mark.parametrize('lp', [64, 512])
def test_frequency_sum_of_a_run_as_a_nonuniform_fft(k, lp, W, dtype, cdtype, bar):
    """W = 8 in float32 arithmetic (float32 data) and W = 14 in float64 (float64 data: ps_nufft_kernel<double>, PnCfg<double>)"""
    rng = np.random.default_rng(k + lp)
    nt, dt, v, tnum, piece = 2048, 1e-8, 1.69e8, 2048, 3
    ws = 2 * np.pi * np.fft.fftfreq(nt, d=dt)
    kx = 2 * np.pi * np.fft.fftfreq(tnum, d=1.0)
    w = np.abs(ws[1:nt // 2 + 1])
    c = 0.5 * v * kx[k]
    w = w[w * w > c * c]
    phi = w * dt * np.sqrt(1 - (c / w) ** 2)                           # phase per depth step (:411-415)
    coef = rng.standard_normal(len(w)) + 1j * rng.standard_normal(len(w))
    n = np.arange(lp)
    want = (coef[None, :] * np.exp(1j * np.outer(piece * lp + n + 1, phi))).sum(1)      # the direct sum, float64
    g = SIGMA * lp
    d = (coef * np.exp(1j * phi * (piece * lp + 1 + lp / 2))).astype(cdtype)            # coefficients at the middle of the piece
    u = phi * g / (2 * np.pi)
    m0 = np.floor(u).astype(int)
    fr = (u - m0).astype(dtype)
    grid = np.zeros(g, dtype=cdtype)
    for dm in range(-W // 2 + 1, W // 2 + 1):                          # spreading (the kernel gathers; the sums are the same)
        np.add.at(grid, (m0 + dm) % g, (d * window(fr - dtype(dm), dtype, W)).astype(cdtype))
    ghat = (np.fft.ifft(grid) * g).astype(cdtype)
    npr = np.arange(-lp // 2, lp // 2)
    got = ghat[npr % g] * correction(lp, W)
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert err < bar, err


@pytest.mark.parametrize('W,dtype,cdtype,bar', [(8, np.float32, np.complex64, 1.5e-6), (14, np.float64, np.complex128, 5e-12)])
@pytest.mark.parametrize('k', [1, 37, 700])
@pytest.mark.parametrize('lp', [16, 64, 512])
def test_a_pair_of_wavenumbers_as_one_transform_with_mirrored_nodes(k, lp, W, dtype, cdtype, bar):
    """Round 6 (ps_nufft_kernel<T, true>): only Re ifft_k TK is kept (mig_python.py:282), so of rows k and -k only
    G = (TK[k] + conj TK[-k]) / 2 = 1/2 sum_w [a_w e^{+i phi_w (n+1)} + conj(b_w) e^{-i phi_w (n+1)}] is needed (the phases depend
    on kx^2): ONE transform whose nodes are +-phi_w.  The positive nodes live in the grid's half [0, G/2], the mirrored ones in the
    other; a grid point m and its mirror image -m share the window value psi(u_w - m).  The Nyquist row (w < 0: phase -phi_N per step)
    is the mirrored partner's node at +|phi_N| with the two coefficients' roles swapped.  lp = 16 with W = 14: the smallest grid (32
    points), where the two index sets -W/2 .. G/2 + W/2 and their mirror images overlap most."""
    rng = np.random.default_rng(k + lp)
    nt, dt, v, tnum, piece = 2048, 1e-8, 1.69e8, 2048, 1
    ws = 2 * np.pi * np.fft.fftfreq(nt, d=dt)
    kx = 2 * np.pi * np.fft.fftfreq(tnum, d=1.0)
    wall = np.abs(ws[1:nt // 2 + 1])                                   # ... the last one is the Nyquist row's |w|
    c = 0.5 * v * kx[k]
    assert kx[tnum - k] == -kx[k]
    alive = wall * wall > c * c
    w = wall[alive]
    phi = w * dt * np.sqrt(1 - (c / w) ** 2)
    sign = np.ones(len(w))
    sign[-1] = -1.0                                                    # the Nyquist row walks backwards (w = -pi / dt)
    a = rng.standard_normal(len(w)) + 1j * rng.standard_normal(len(w))             # row k
    b = rng.standard_normal(len(w)) + 1j * rng.standard_normal(len(w))             # row tnum - k
    n = np.arange(lp)
    steps = piece * lp + n + 1
    tk_k = (a[None, :] * np.exp(1j * np.outer(steps, sign * phi))).sum(1)
    tk_m = (b[None, :] * np.exp(1j * np.outer(steps, sign * phi))).sum(1)
    want = 0.5 * (tk_k + np.conj(tk_m))
    g = SIGMA * lp
    mid = piece * lp + 1 + lp / 2
    da = a * np.exp(1j * sign * phi * mid)                             # row k's coefficient: node at sign * phi
    db = np.conj(b * np.exp(1j * sign * phi * mid))                    # the partner's, mirrored: node at -sign * phi
    # every node by |phi|: D goes to the grid's positive half, D2 to its mirror image (the Nyquist row trades places)
    D = np.where(sign > 0, da, db).astype(cdtype)
    D2 = np.where(sign > 0, db, da).astype(cdtype)
    u = phi * g / (2 * np.pi)
    m0 = np.floor(u).astype(int)
    fr = (u - m0).astype(dtype)
    grid = np.zeros(g, dtype=cdtype)
    for dm in range(-W // 2 + 1, W // 2 + 1):
        wgt = window(fr - dtype(dm), dtype, W)                         # psi(u - m), m = m0 + dm: serves D -> g[m] and D2 -> g[-m]
        np.add.at(grid, (m0 + dm) % g, (D * wgt).astype(cdtype))
        np.add.at(grid, (-(m0 + dm)) % g, (D2 * wgt).astype(cdtype))
    ghat = (np.fft.ifft(grid) * g).astype(cdtype)
    npr = np.arange(-lp // 2, lp // 2)
    got = 0.5 * ghat[npr % g] * correction(lp, W)
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert err < bar, err
    # and the direct single steps of a smeared boundary: f1 e^{ip} + conj(f2) e^{-ip} from (f1 + f2) and (f1 - f2) -- one sum's cost
    p = np.outer(steps, sign * phi)
    s, q = a + b, a - b
    direct = 0.5 * ((s.real * np.cos(p) - s.imag * np.sin(p)) + 1j * (q.real * np.sin(p) + q.imag * np.cos(p))).sum(1)
    assert np.linalg.norm(direct - want) / np.linalg.norm(want) < 1e-12


def test_the_identities_the_half_layout_and_the_hermitian_inverse_rest_on():
    """(1) The spectrum of a REAL radargram, transform over the traces first and only k = 0 .. tnum/2 kept (ps_run's P.fhalf): the rows
    k > tnum/2 of fft2 are the mirrored frequencies of row tnum - k, conjugated -- FK[tnum - k][w] = conj FK[k][-w] -- which is how a
    pair of wavenumbers reads both of its rows out of one (ps_load_slot_k).  (2) Only the real part of the inverse transform over the
    wavenumbers is kept (mig_python.py:282): Re ifft_k TK = irfft_k G with G[k] = (TK[k] + conj TK[tnum - k]) / 2, k = 0 .. tnum/2
    (ps_transpose_herm + a real row transform of half the length), for ANY complex TK."""
    rng = np.random.default_rng(3)
    snum, tnum, nt = 24, 16, 32
    x = rng.standard_normal((snum, tnum))
    full = np.fft.fft2(x, (nt, tnum))                                  # [w][k], as mig_python.py:270
    half = np.fft.fft(np.fft.rfft(x, axis=1), n=nt, axis=0)            # R2C over the traces, zero-padded transform over time: [w][k <= tnum/2]
    assert np.allclose(half, full[:, :tnum // 2 + 1], atol=1e-12)
    for k in range(tnum // 2 + 1, tnum):
        mirrored = np.conj(half[(-np.arange(nt)) % nt, tnum - k])
        assert np.allclose(mirrored, full[:, k], atol=1e-12)
    tk = rng.standard_normal((snum, tnum)) + 1j * rng.standard_normal((snum, tnum))      # [tau][k], any complex array
    g = 0.5 * (tk + np.conj(tk[:, (-np.arange(tnum)) % tnum]))
    assert np.allclose(np.fft.irfft(g[:, :tnum // 2 + 1], n=tnum, axis=1), np.fft.ifft(tk, axis=1).real, atol=1e-13)

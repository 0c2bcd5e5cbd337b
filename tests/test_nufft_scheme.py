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

"""Parity of the HIP phase-shift (Gazdag) path and the taper-only T-K stub
with the reference's golden vectors and the CPU oracle.

Stated tolerances: float64 data  max|diff| <= 1e-10 * max|ref|  (the reference's
                   multiplicative recurrence and this kernel's differ by
                   rounding only; observed <= 5.2e-12 over 1300 randomized cases,
                   profiles/r04_fuzz.txt; 1e-9 until round 4)
                   float32 data  relative L2 <= 2e-4."""
import os

import numpy as np
import pytest

from conftest import ROOT, golden, golden_names, make_dat, rel_l2, rel_max

pytestmark = pytest.mark.gpu

F64_TOL = 1e-10
F32_L2 = 2e-4


@pytest.mark.parametrize('name', golden_names('P1') + golden_names('P2') + golden_names('P5'))
def test_golden(hip, name):
    g = golden(name)
    dat = make_dat(g)
    vel = float(g['vel']) if g['vel'].ndim == 0 else g['vel']
    from impdar_amd.lib import migrationlib
    migrationlib.migrationPhaseShift(dat, vel=vel, htaper=int(g['htaper']), vtaper=int(g['vtaper']))
    assert dat.data.dtype == np.float64 and dat.data.shape == g['expected'].shape
    assert rel_max(dat.data, g['expected']) < F64_TOL, rel_max(dat.data, g['expected'])


@pytest.mark.parametrize('name', ['P1r_phsh_const_ricker', 'P2_phsh_vz_64x48', 'P5_phsh_vz_boundary_100x64'])
def test_golden_float32(hip, name):
    g = golden(name)
    dat = make_dat(g)
    dat.data = dat.data.astype(np.float32)
    vel = float(g['vel']) if g['vel'].ndim == 0 else g['vel']
    from impdar_amd.lib import migrationlib
    migrationlib.migrationPhaseShift(dat, vel=vel, htaper=int(g['htaper']), vtaper=int(g['vtaper']))
    assert dat.data.dtype == np.float64
    assert rel_l2(dat.data, g['expected']) < F32_L2, rel_l2(dat.data, g['expected'])


@pytest.mark.parametrize('name', golden_names('P4'))
def test_golden_ffd(hip, name):
    """2-D v(x,z) Fourier finite-difference branch (mig_python.py:428-432,448-487,496-540) with the
    reference's own 3-column table (test/input_data/velocity_lateral.txt, stored in the fixture)."""
    g = golden(name)
    dat = make_dat(g)
    from impdar_amd.lib import migrationlib
    migrationlib.migrationPhaseShift(dat, vel=g['vel'], htaper=int(g['htaper']), vtaper=int(g['vtaper']))
    assert dat.data.dtype == np.float64 and dat.data.shape == g['expected'].shape
    assert rel_max(dat.data, g['expected']) < F64_TOL, rel_max(dat.data, g['expected'])


def test_ffd_vs_oracle_and_float32(hip, tmp_path):
    """A second geometry against the oracle, the table read from a file (test_PhaseShiftLateral,
    test/test_migrationlib.py:133-135), and float32 data (computed in float64 here)."""
    from oracle import mig_oracle
    from impdar_amd.lib import migrationlib
    g = golden('P4_phsh_ffd_32x16')
    rng = np.random.default_rng(11)
    snum, tnum = 32, 19
    data = rng.standard_normal((snum, tnum))
    gg = dict(data=data, travel_time=g['travel_time'], dist=np.arange(tnum) * 5.0 / 1e3,
              trace_int=np.ones(tnum) * 5.0, dt=g['dt'])
    want = mig_oracle.phase_shift(data, float(g['dt']), gg['trace_int'], gg['travel_time'], gg['dist'], g['vel'], 4, 3)
    fn = tmp_path / 'velocity_lateral.txt'
    np.savetxt(fn, g['vel'])
    dat = make_dat(gg)
    migrationlib.migrationPhaseShift(dat, vel_fn=str(fn), htaper=4, vtaper=3)
    assert rel_max(dat.data, want) < F64_TOL, rel_max(dat.data, want)
    dat = make_dat(gg)
    dat.data = dat.data.astype(np.float32)
    migrationlib.migrationPhaseShift(dat, vel=g['vel'], htaper=4, vtaper=3)
    assert dat.data.dtype == np.float64
    assert rel_l2(dat.data, want) < F32_L2, rel_l2(dat.data, want)


@pytest.mark.parametrize('tnum', [64, 256])
def test_ffd_chain_in_one_workgroup(hip, tnum, monkeypatch):
    """Power-of-two trace counts run the whole (tau, omega) chain in one persistent workgroup (its own LDS
    transforms); every other count, and IMPDAR_FFD_CHAIN=0, the launch-per-step form with rocFFT.  Both against the
    oracle, and against each other."""
    from oracle import mig_oracle
    from impdar_amd.lib import migrationlib
    g = golden('P4_phsh_ffd_32x16')
    rng = np.random.default_rng(tnum)
    snum = 32
    data = rng.standard_normal((snum, tnum))
    gg = dict(data=data, travel_time=g['travel_time'], dist=np.arange(tnum) * 5.0 / 1e3,
              trace_int=np.ones(tnum) * 5.0, dt=g['dt'])
    want = mig_oracle.phase_shift(data, float(g['dt']), gg['trace_int'], gg['travel_time'], gg['dist'], g['vel'], 4, 3)
    assert np.isfinite(want).all()
    outs = {}
    for chain in ('1', '0'):
        monkeypatch.setenv('IMPDAR_FFD_CHAIN', chain)
        dat = make_dat(gg)
        migrationlib.migrationPhaseShift(dat, vel=g['vel'], htaper=4, vtaper=3)
        outs[chain] = dat.data
        assert rel_max(dat.data, want) < F64_TOL, (chain, rel_max(dat.data, want))
    assert rel_max(outs['1'], outs['0']) < F64_TOL


@pytest.mark.parametrize('kind', ['vz', 'const'])
def test_config5_size_spot_wavenumbers_and_linearity(hip, kind):
    """BASELINE config 5 (8192 x 8192 float32, 1-D v(z) table; also constant v) at full size.  Wavenumbers are independent
    in phaseShift (mig_python.py:438-487), so the oracle is run on a few (k, -k) column pairs of the 2-D
    spectrum and compared with the same columns of the GPU image transformed back over x:
    fft_x(Re ifft_k TK)[k] = (TK[k] + conj(TK[-k])) / 2.  Plus linearity of the whole image."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from oracle import mig_oracle
    n = 8192
    geo = synth.geometry(n, n)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    tab = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]]) if kind == 'vz' else 1.69e8
    rng = np.random.default_rng(2)
    x = rng.standard_normal((n, n)).astype(np.float32)
    y = rng.standard_normal((n, n)).astype(np.float32)

    def run(a):
        d = RadarData(None)
        d.data, d.snum, d.tnum = a, n, n
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        d.migrate('phsh', vel=tab, htaper=100, vtaper=1000)
        return d.data

    mx = run(x.copy())
    assert mx.dtype == np.float64 and mx.shape == (n, n) and np.isfinite(mx).all()
    if kind == 'vz':
        my = run(y.copy())
        mz = run((2 * x - 3 * y).astype(np.float32))
        assert rel_l2(mz, 2.0 * mx - 3.0 * my) < F32_L2
        del my, mz
    # oracle on spot wavenumbers (the taper and the 2-D FFT restated with NumPy, float32 data as given)
    tap = mig_oracle._apply_taper(x, 100, 1000, inplace_form=True).astype(np.float32)
    # 200 and 4000: wavenumbers that hold a frequency exactly on the evanescent boundary of the first layer
    # (0.5 v dt / dx = 0.845 = 169/200: wavenumber 200 q meets frequency 169 q) -- the fp64 boundary walk at full size;
    # 4096: the wavenumber Nyquist row (its own mirror image)
    ks = np.array([0, 1, 37, 200, 1000, 4000, 4095, 4096])
    kxa = mig_oracle._kx(n, geo['trace_int'], geo['dist'])
    wsa = 2. * np.pi * np.fft.fftfreq(n, d=geo['dt'])
    assert abs(1. - (0.5 * 1.69e8 * kxa[200] / wsa[169]) ** 2) < 1e-8 and abs(1. - (0.5 * 1.69e8 * kxa[4000] / wsa[3380]) ** 2) < 1e-8
    cols = np.concatenate([ks, (n - ks) % n])
    FKc = np.fft.fft(np.fft.fft(tap.astype(np.float64), axis=1)[:, cols], n=n, axis=0)     # (nt = n, len(cols))
    kx = mig_oracle._kx(n, geo['trace_int'], geo['dist'])[cols]
    ws = 2. * np.pi * np.fft.fftfreq(n, d=geo['dt'])
    vmig = mig_oracle.get_velocity_profile(geo['travel_time'], tab)
    TK = mig_oracle.phase_shift_tk(FKc, vmig, kx, ws, geo['dt'], geo['travel_time'], n, len(cols))
    want = 0.5 * (TK[:, :len(ks)] + np.conj(TK[:, len(ks):]))
    got = np.fft.fft(mx, axis=1)[:, ks]
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    print('config 5 (%s) spot-wavenumber relative L2 error %.3g' % (kind, err))
    assert err < F32_L2, err


_SPOT_KS = np.array([0, 1, 37, 200, 1000, 4000, 4095, 4096])
_spot_cache = {}


def _spot_oracle(n, geo, x64, vmig):
    """phaseShift (mig_python.py:438-487) on the (k, -k) column pairs of _SPOT_KS of the 2-D spectrum of the tapered
    float64 radargram -> what fft_x of the migrated image must hold in those columns,
    fft_x(Re ifft_k TK)[k] = (TK[k] + conj(TK[-k])) / 2  (wavenumbers are independent in phaseShift)."""
    from oracle import mig_oracle
    ks = _SPOT_KS
    tap = mig_oracle._apply_taper(x64, 100, 1000, inplace_form=True)
    cols = np.concatenate([ks, (n - ks) % n])
    FKc = np.fft.fft(np.fft.fft(tap, axis=1)[:, cols], n=n, axis=0)
    del tap
    kx = mig_oracle._kx(n, geo['trace_int'], geo['dist'])[cols]
    ws = 2. * np.pi * np.fft.fftfreq(n, d=geo['dt'])
    TK = mig_oracle.phase_shift_tk(FKc, vmig, kx, ws, geo['dt'], geo['travel_time'], n, len(cols))     # ~30 s
    return 0.5 * (TK[:, :len(ks)] + np.conj(TK[:, len(ks):]))


def _config5_profile(name, n, geo):
    from oracle import mig_oracle
    u = np.linspace(0., 1., n)
    if name == 'gradient':
        return np.ascontiguousarray(1.69e8 + 0.5e8 * u)
    if name == 'wavy':
        return np.ascontiguousarray(1.8e8 + 0.15e8 * np.sin(7. * u) + 0.1e8 * u)
    if name == 'falling':
        return np.ascontiguousarray(2.2e8 - 0.5e8 * u)
    if name == 'firn':                           # fast near the surface, flat below: what a firn column looks like
        return np.ascontiguousarray(1.69e8 + 0.6e8 * np.exp(-np.arange(n) * geo['dt'] / 0.8e-6))
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    if name == 'vz4':                            # BASELINE config 5's table
        tab = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
    else:
        assert name == 'layers41'
        tab = np.stack([np.linspace(1.69e8, 2.2e8, 41), np.linspace(0., 2.0 * Rp, 41)], axis=1)
    return np.ascontiguousarray(mig_oracle.get_velocity_profile(geo['travel_time'], tab))


# what runs by itself at 8192 x 8192 (the library's estimate decides between the series path and the per-step / runs kernels:
# phaseshift.hip, ps_series_plan.h)
_CONFIG5_KERNEL = {('gradient', 'float32'): 'ps_series_kernel', ('gradient', 'float64'): 'ps_smooth_kernel',
                   ('wavy', 'float32'): 'ps_series_kernel', ('wavy', 'float64'): 'ps_smooth_kernel',
                   ('falling', 'float32'): 'ps_series_kernel', ('falling', 'float64'): 'ps_smooth_kernel',
                   ('firn', 'float32'): 'ps_series_kernel', ('firn', 'float64'): 'ps_series_kernel',
                   ('vz4', 'float64'): 'ps_nufft_kernel', ('const', 'float64'): 'ps_nufft_kernel',
                   ('layers41', 'float32'): 'ps_nufft_kernel', ('layers41', 'float64'): 'ps_nufft_kernel'}


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('profile', ['gradient', 'wavy', 'layers41', 'falling', 'firn', 'vz4', 'const'])
def test_config5_size_spot_wavenumbers_per_step_profiles_and_many_layers(hip, profile, dtype):
    """Round 6: the series path (ps_series_kernel: a falling gradient, a firn column) and the float64 transform path with its
    first-order term (config 5's table; a constant velocity) at the size they are quoted at, asserting WHICH kernel ran.
    The kernels of round 4 at the size they are quoted at (8192 x 8192): ps_smooth32_kernel / ps_smooth_kernel carry
    sqrt(coss) and the rotation by Newton / series updates between float64 anchors and expand about one velocity per 32
    steps -- a scheme whose error could grow with the number of steps -- and a 41-row table (40 layers of ~200 steps)
    is what the many-run paths get.  Spot wavenumbers (zero, low, two that hold a frequency on the evanescent boundary of
    1.69e8 m/s, high, the Nyquist row) against the oracle's literal per-step loop (mig_python.py:438-487)."""
    import ctypes as C
    import json
    from impdar_amd import _hip, synth
    from oracle import mig_oracle
    lib, ctx = _hip.load(), _hip.context()
    n = 8192
    geo = synth.geometry(n, n)
    if profile in ('vz4', 'const') and dtype == np.float32:
        pytest.skip('test_config5_size_spot_wavenumbers_and_linearity holds the float32 table and constant velocity')
    vm = _config5_profile(profile, n, geo) if profile != 'const' else None
    x = np.random.default_rng(2).standard_normal((n, n)).astype(np.float32)
    if profile not in _spot_cache:
        _spot_cache.clear()                      # (one 8192 x 8 complex array at a time is plenty)
        _spot_cache[profile] = _spot_oracle(n, geo, x.astype(np.float64), vm if vm is not None else 1.69e8)
    want = _spot_cache[profile]
    data = x.astype(dtype)
    del x
    kx = mig_oracle._kx(n, geo['trace_int'], geo['dist'])
    ws = 2. * np.pi * np.fft.fftfreq(n, d=geo['dt'])
    out = np.empty((n, n), dtype=dtype)
    tt = np.ascontiguousarray(geo['travel_time'], dtype=np.float64)
    dp = C.POINTER(C.c_double)
    _hip.check(lib.impdar_phaseshift(ctx, data.ctypes.data_as(C.c_void_p), _hip.dtype_code(dtype), n, n, n,
                                     kx.ctypes.data_as(dp), ws.ctypes.data_as(dp), C.c_double(geo['dt']),
                                     tt.ctypes.data_as(dp), C.c_double(1.69e8 if vm is None else 0.0),
                                     vm.ctypes.data_as(dp) if vm is not None else None, n if vm is not None else 0, C.c_double(100.),
                                     C.c_double(1000.), out.ctypes.data_as(C.c_void_p)), 'impdar_phaseshift')
    buf = C.create_string_buffer(1024)
    _hip.check(lib.impdar_ctx_last_metrics(ctx, buf, len(buf)), 'metrics')
    kernel = json.loads(buf.value.decode())['kernel']
    want_kernel = _CONFIG5_KERNEL.get((profile, np.dtype(dtype).name))
    if want_kernel:
        assert kernel == want_kernel, kernel
    del data
    assert np.isfinite(out).all()
    got = np.fft.fft(out.astype(np.float64), axis=1)[:, _SPOT_KS]
    err_l2 = np.linalg.norm(got - want) / np.linalg.norm(want)
    err_max = np.max(np.abs(got - want)) / np.max(np.abs(want))
    print('8192^2 %s %s (%s): spot-wavenumber rel L2 %.3g, rel max %.3g' % (profile, np.dtype(dtype).name, kernel, err_l2, err_max))
    if dtype == np.float32:
        assert err_l2 < F32_L2, err_l2
    else:
        assert err_max < F64_TOL, err_max


@pytest.mark.parametrize('snum,tnum', [(300, 64), (257, 65), (1100, 24), (40, 7)])
@pytest.mark.parametrize('layered', [False, True])
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_hermitian_walk_against_the_full_walk_and_the_oracle(hip, monkeypatch, snum, tnum, layered, dtype):
    """The radargram is real, so the default walk covers the Nyquist row and frequencies 1 .. nt/2-1 (doubled) and adds
    the zero-frequency row at kx = 0 (mig_python.py:268-270, 282: FK Hermitian, only ifft(TK).real kept);
    IMPDAR_PS_HERMITIAN=0 walks all nt two-sided frequencies as the reference does.  Both against the oracle at the
    stated bars and against each other; even and odd trace counts (with and without a wavenumber Nyquist row), a
    radargram with a large mean (the zero-frequency row carries it)."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from impdar_amd.lib import migrationlib
    from oracle import mig_oracle
    geo = synth.geometry(snum, tnum)
    data = (synth.noise_radargram(snum, tnum, seed=snum + tnum) + 3.0).astype(dtype)
    if layered:
        Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
        vel = np.array([[1.68e8, 0.], [1.68e8, 0.3 * Rp], [1.8e8, 0.6 * Rp], [1.9e8, 1.2 * Rp]])
    else:
        vel = 1.69e8
    want = mig_oracle.phase_shift(data.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'],
                                  geo['dist'], vel, 5, 7)
    tol, measure = (F64_TOL, rel_max) if dtype == np.float64 else (F32_L2, rel_l2)
    outs = {}
    for herm in ('1', '0'):
        monkeypatch.setenv('IMPDAR_PS_HERMITIAN', herm)
        d = RadarData(None)
        d.data, d.snum, d.tnum = data.copy(), snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        migrationlib.migrationPhaseShift(d, vel=vel, htaper=5, vtaper=7)
        outs[herm] = d.data
        assert measure(d.data, want) < tol, (herm, measure(d.data, want))
    assert measure(outs['1'], outs['0']) < tol


@pytest.mark.parametrize('snum,tnum', [(520, 40), (1000, 33), (2100, 16)])
@pytest.mark.parametrize('kind', ['const', 'layers', 'uneven_layers', 'boundary'])
def test_matrix_core_path_against_the_vector_kernels_and_the_oracle(hip, monkeypatch, snum, tnum, kind):
    """float32 data with a few long runs of constant velocity: the frequency sums run on the matrix cores
    (ps_mfma_kernel: states x step factors per 16-step tile, float16 hi/lo operands, float32 accumulation) and the
    boundary frequencies in ps_edge_kernel.  Held to the oracle at the float32 bar, compared with the vector kernels
    (IMPDAR_PS_MFMA=0), with and without the half walk: sizes whose last tile / row block is ragged, runs that start
    anywhere, a first layer at 1.68e8 m/s that puts frequencies exactly on the evanescent boundary."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from impdar_amd.lib import migrationlib
    from oracle import mig_oracle
    geo = synth.geometry(snum, tnum)
    data = (synth.noise_radargram(snum, tnum, seed=snum) + 0.5).astype(np.float32)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    vel = {'const': 1.69e8,
           'layers': np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]]),
           'uneven_layers': np.array([[1.6e8, 0.], [1.6e8, 0.07 * Rp], [1.75e8, 0.61 * Rp], [1.82e8, 0.83 * Rp], [1.9e8, 1.2 * Rp]]),
           'boundary': np.array([[1.68e8, 0.], [1.68e8, 0.45 * Rp], [1.8e8, 0.7 * Rp], [1.9e8, 1.2 * Rp]])}[kind]
    want = mig_oracle.phase_shift(data.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'],
                                  geo['dist'], vel, 20, 30)
    import ctypes as C
    import json
    from impdar_amd import _hip
    outs, kernels = {}, {}
    # '2': ps_mfma_kernel; '6': ps_nufft_kernel -- the frequency sum of a run as a non-uniform FFT (what a call runs by itself on
    # such tables); '7': ps_series_kernel (round 6; constant velocity: not its call)
    for mfma, herm in (('2', '1'), ('0', '1'), ('2', '0'), ('6', '1'), ('1', '1'), ('7', '1')):
        monkeypatch.setenv('IMPDAR_PS_MFMA', mfma)
        monkeypatch.setenv('IMPDAR_PS_HERMITIAN', herm)
        d = RadarData(None)
        d.data, d.snum, d.tnum = data.copy(), snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        migrationlib.migrationPhaseShift(d, vel=vel, htaper=20, vtaper=30)
        outs[mfma, herm] = d.data
        assert rel_l2(d.data, want) < F32_L2, (mfma, herm, rel_l2(d.data, want))
        buf = C.create_string_buffer(1024)
        _hip.check(_hip.load().impdar_ctx_last_metrics(_hip.context(), buf, len(buf)), 'metrics')
        kernels[mfma, herm] = json.loads(buf.value.decode())['kernel']
    print('%s %dx%d: rel L2 vs oracle: ps_mfma_kernel %.2e, vector kernels %.2e, on the full walk %.2e, ps_nufft_kernel %.2e, mode 7 (%s) %.2e'
          % (kind, snum, tnum, rel_l2(outs['2', '1'], want), rel_l2(outs['0', '1'], want), rel_l2(outs['2', '0'], want),
             rel_l2(outs['6', '1'], want), kernels['7', '1'], rel_l2(outs['7', '1'], want)))
    for herm in ('1', '0'):      # (short records with layers: rows mostly padding, it declines)
        assert kernels['2', herm] in ('ps_mfma_kernel', kernels['0', '1']), kernels
    assert 'mfma' not in kernels['0', '1'], kernels
    assert kernels['6', '1'] == kernels['1', '1'] == 'ps_nufft_kernel', kernels
    assert kernels['7', '1'] == ('ps_series_kernel' if kind != 'const' else kernels['0', '1']), kernels
    assert np.array_equal(outs['6', '1'], outs['1', '1'])
    for m in ('2', '6', '7'):
        assert rel_l2(outs[m, '1'], outs['0', '1']) < F32_L2
        # the matrix-core result must not be worse than a few times the vector kernels' own float32 error
        assert rel_l2(outs[m, '1'], want) < max(5.0 * rel_l2(outs['0', '1'], want), 2e-6), (m, rel_l2(outs[m, '1'], want))


@pytest.mark.parametrize('snum,tnum', [(520, 33), (700, 24), (1100, 40), (2100, 16), (4200, 6)])
@pytest.mark.parametrize('kind', ['layers13', 'layers40', 'uneven', 'boundary', 'slowing', 'thick'])
def test_many_runs_matrix_core_path_against_the_vector_kernels_and_the_oracle(hip, monkeypatch, snum, tnum, kind):
    """float32 data with MANY layers of constant velocity (what getVelocityProfile, mig_python.py:582-604, makes of a table
    of N rows: N - 1 layers, every boundary smeared over single steps): ps_runs_kernel -- 8-step tiles against float32
    matrix-core products, phases generated in the kernel, single steps as rows of their own, partial images per 1024
    frequencies (4200 samples: 4 parts).  Held to the oracle at the float32 bar; against the vector kernels
    (IMPDAR_PS_MFMA=0); with the reference's walk over all frequencies; the metrics line names the kernel.  Tables: equal
    layers, layers of very different thickness (long runs are cut into pieces of <= 512 steps), a first layer at 1.68e8
    m/s that puts frequencies exactly on the evanescent boundary, a velocity that FALLS with depth (frequencies once
    evanescent stay out: :484-485), two thick layers (IMPDAR_PS_MFMA=3 asks for this kernel where ps_mfma_kernel would
    take the call)."""
    import ctypes as C
    import json
    from impdar_amd import _hip, synth
    from impdar_amd.lib.RadarData import RadarData
    from impdar_amd.lib import migrationlib
    from oracle import mig_oracle
    geo = synth.geometry(snum, tnum)
    data = (synth.noise_radargram(snum, tnum, seed=snum + 1) + 0.5).astype(np.float32)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    lin = lambda n, v0, v1: np.stack([np.linspace(v0, v1, n), np.linspace(0., 1.3 * Rp, n)], axis=1)
    vel = {'layers13': lin(13, 1.69e8, 2.1e8), 'layers40': lin(40, 1.68e8, 1.9e8),
           'uneven': np.array([[1.6e8, 0.], [1.6e8, 0.02 * Rp], [1.66e8, 0.05 * Rp], [1.75e8, 0.61 * Rp], [1.78e8, 0.63 * Rp],
                               [1.82e8, 0.83 * Rp], [1.86e8, 0.9 * Rp], [1.9e8, 1.3 * Rp]]),
           'boundary': np.concatenate([[[1.68e8, 0.]], lin(9, 1.68e8, 1.95e8) + [[0., 0.3 * Rp]] * 9])[:, :],
           'slowing': lin(11, 2.1e8, 1.7e8),
           'thick': np.array([[1.69e8, 0.], [1.69e8, 0.5 * Rp], [1.9e8, 1.3 * Rp]])}[kind]
    want = mig_oracle.phase_shift(data.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'],
                                  geo['dist'], vel, 20, 30)
    outs, kernels = {}, {}
    for mfma, herm in (('3', '1'), ('0', '1'), ('3', '0')):
        monkeypatch.setenv('IMPDAR_PS_MFMA', mfma)
        monkeypatch.setenv('IMPDAR_PS_HERMITIAN', herm)
        d = RadarData(None)
        d.data, d.snum, d.tnum = data.copy(), snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        migrationlib.migrationPhaseShift(d, vel=vel, htaper=20, vtaper=30)
        outs[mfma, herm] = d.data
        buf = C.create_string_buffer(1024)
        _hip.check(_hip.load().impdar_ctx_last_metrics(_hip.context(), buf, len(buf)), 'metrics')
        kernels[mfma, herm] = json.loads(buf.value.decode())['kernel']
        assert rel_l2(d.data, want) < F32_L2, (mfma, herm, kernels[mfma, herm], rel_l2(d.data, want))
    print('%s %dx%d: rel L2 vs oracle: %s %.2e, %s %.2e, full walk %.2e'
          % (kind, snum, tnum, kernels['3', '1'], rel_l2(outs['3', '1'], want), kernels['0', '1'], rel_l2(outs['0', '1'], want),
             rel_l2(outs['3', '0'], want)))
    assert kernels['3', '1'] == 'ps_runs_kernel' and kernels['3', '0'] == 'ps_runs_kernel' and kernels['0', '1'] != 'ps_runs_kernel', kernels
    assert rel_l2(outs['3', '1'], outs['0', '1']) < F32_L2
    # the matrix-core result must not be worse than a few times the vector kernels' own float32 error
    assert rel_l2(outs['3', '1'], want) < max(5.0 * rel_l2(outs['0', '1'], want), 3e-6)
    # by itself the library picks one of the two matrix-core kernels (this one for more than 16 long runs)
    monkeypatch.delenv('IMPDAR_PS_MFMA')
    monkeypatch.delenv('IMPDAR_PS_HERMITIAN')
    d = RadarData(None)
    d.data, d.snum, d.tnum = data.copy(), snum, tnum
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    migrationlib.migrationPhaseShift(d, vel=vel, htaper=20, vtaper=30)
    buf = C.create_string_buffer(1024)
    _hip.check(_hip.load().impdar_ctx_last_metrics(_hip.context(), buf, len(buf)), 'metrics')
    chosen = json.loads(buf.value.decode())['kernel']
    # (up to 16 long runs: ps_mfma_kernel where its 2048-step row blocks are not mostly padding, i.e. on long records)
    assert chosen in ('ps_runs_kernel', 'ps_mfma_kernel', 'ps_nufft_kernel', 'ps_series_kernel'), (kind, chosen)
    assert chosen == 'ps_runs_kernel' or kind not in ('layers40',), (kind, chosen)
    if chosen == 'ps_runs_kernel':      # (the same sums; the transforms around them may be the library's own or rocFFT's by now)
        assert rel_l2(d.data, outs['3', '1']) < 5e-6


def test_matrix_core_path_leaves_many_short_runs_to_the_vector_kernels(hip, monkeypatch):
    """A table of 40 thin layers (ps_mfma_kernel's rows of 32 tiles would be mostly padding: ps_runs_kernel takes it since
    round 5, the vector kernels with IMPDAR_PS_MFMA=0); the answer is the same either way."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from impdar_amd.lib import migrationlib
    from oracle import mig_oracle
    snum, tnum = 700, 24
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=3).astype(np.float32)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    vel = np.stack([np.linspace(1.68e8, 1.9e8, 40), np.linspace(0., 1.2 * Rp, 40)], axis=1)
    want = mig_oracle.phase_shift(data.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'],
                                  geo['dist'], vel, 20, 30)
    for mfma in ('1', '0'):
        monkeypatch.setenv('IMPDAR_PS_MFMA', mfma)
        d = RadarData(None)
        d.data, d.snum, d.tnum = data.copy(), snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        migrationlib.migrationPhaseShift(d, vel=vel, htaper=20, vtaper=30)
        assert rel_l2(d.data, want) < F32_L2


@pytest.mark.parametrize('snum,tnum,dx', [(200, 512, 1.0), (700, 256, 0.5)])
def test_transform_path_keeps_the_boundary_frequencies_of_a_constant_velocity(hip, monkeypatch, snum, tnum, dx):
    """Round numbers (2e8 m/s, 5 ns, 0.5 / 1 m) put frequencies exactly ON the evanescent boundary v kx / 2 = w.  The
    reference keeps a frequency when vkx2 < w^2 in ITS arithmetic (mig_python.py:411-412) -- with a phase of ~1e-8 w dt, i.e. as a
    constant term of every depth step.  ps_nufft_kernel first decided it from 1 - (v kx / 2w)^2 with 1/w rounded separately and
    dropped such frequencies: 1.3e-2 / 6.6e-3 of the image on exactly these two geometries (fuzz, profiles/r05_fuzz.txt)."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from impdar_amd.lib import migrationlib
    from oracle import mig_oracle
    geo = synth.geometry(snum, tnum, dt=5e-9, dx=dx)
    data = synth.noise_radargram(snum, tnum, seed=snum + tnum).astype(np.float32)
    want = mig_oracle.phase_shift(data.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'], geo['dist'], 2.0e8, 7, 9)
    for mode in ('6', '2', '0'):
        monkeypatch.setenv('IMPDAR_PS_MFMA', mode)
        d = RadarData(None)
        d.data, d.snum, d.tnum = data.copy(), snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        migrationlib.migrationPhaseShift(d, vel=2.0e8, htaper=7, vtaper=9)
        assert rel_l2(d.data, want) < 1e-5, (mode, rel_l2(d.data, want))


def test_matrix_core_path_hands_over_when_a_wavenumber_has_more_boundary_frequencies_than_it_lists(hip, monkeypatch):
    """ps_setup_kernel takes every boundary frequency out of the matrix-core sums and lists the first 16 per wavenumber
    for ps_edge_kernel; beyond that contributions would be missing from TK.  The host reads the counters back after the
    launch and, on an overflow, discards the result and lets the vector kernels produce it (they walk every boundary
    frequency themselves).  Forced here by the test hook on the 'boundary' table (a first layer at 1.68e8 m/s puts
    frequencies exactly on the evanescent boundary): same answer as the vector kernels asked for by name, bit for bit."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from impdar_amd.lib import migrationlib
    from oracle import mig_oracle
    import ctypes as C
    import json
    from impdar_amd import _hip
    snum, tnum = 2100, 16
    geo = synth.geometry(snum, tnum)
    data = (synth.noise_radargram(snum, tnum, seed=snum) + 0.5).astype(np.float32)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    vel = np.array([[1.68e8, 0.], [1.68e8, 0.45 * Rp], [1.8e8, 0.7 * Rp], [1.9e8, 1.2 * Rp]])
    want = mig_oracle.phase_shift(data.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'],
                                  geo['dist'], vel, 20, 30)
    outs, kernels = {}, {}
    monkeypatch.setenv('IMPDAR_PS_FFT', 'own')       # (one transform implementation for the bit-for-bit comparison below)
    for name, env in (('mfma', {}), ('overflow', {'IMPDAR_PS_TEST_EDGE_OVERFLOW': '1'}), ('vector', {'IMPDAR_PS_MFMA': '0'})):
        for k in ('IMPDAR_PS_TEST_EDGE_OVERFLOW', 'IMPDAR_PS_MFMA'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        d = RadarData(None)
        d.data, d.snum, d.tnum = data.copy(), snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        migrationlib.migrationPhaseShift(d, vel=vel, htaper=20, vtaper=30)
        outs[name] = d.data
        assert rel_l2(d.data, want) < F32_L2, (name, rel_l2(d.data, want))
        buf = C.create_string_buffer(1024)
        _hip.check(_hip.load().impdar_ctx_last_metrics(_hip.context(), buf, len(buf)), 'metrics')
        kernels[name] = json.loads(buf.value.decode())['kernel']
    assert kernels['mfma'] == 'ps_nufft_kernel' and kernels['overflow'] == kernels['vector'] and 'nufft' not in kernels['vector'] and 'mfma' not in kernels['vector'], kernels
    assert np.array_equal(outs['overflow'], outs['vector'])


def test_hermitian_walk_is_refused_when_the_axes_are_not_antisymmetric(hip):
    """The C entry point takes kx and ws from the caller; the half walk needs kx[-k] = -kx[k] and ws[-i] = -ws[i].
    With a wavenumber axis that is not (a caller's own, shifted axis) the library must fall back to all nt frequencies:
    checked against the oracle's phase_shift_tk driven with the same axes."""
    import ctypes as C
    from impdar_amd import _hip, synth
    from oracle import mig_oracle
    lib, ctx = _hip.load(), _hip.context()
    snum, tnum = 120, 32
    nt = 128
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=3)
    kx = mig_oracle._kx(tnum, geo['trace_int'], geo['dist']) + 0.013           # not antisymmetric any more
    ws = 2. * np.pi * np.fft.fftfreq(nt, d=geo['dt'])
    out = np.empty((snum, tnum))
    tt = np.ascontiguousarray(geo['travel_time'], dtype=np.float64)
    dp = C.POINTER(C.c_double)
    _hip.check(lib.impdar_phaseshift(ctx, data.ctypes.data_as(C.c_void_p), 1, snum, tnum, nt,
                                     kx.ctypes.data_as(dp), ws.ctypes.data_as(dp), C.c_double(geo['dt']),
                                     tt.ctypes.data_as(dp), C.c_double(1.69e8), None, 0, C.c_double(5.), C.c_double(7.),
                                     out.ctypes.data_as(C.c_void_p)), 'impdar_phaseshift')
    tap = mig_oracle._apply_taper(data, 5, 7, inplace_form=True)
    FK = np.fft.fft2(tap, (nt, tnum))
    TK = mig_oracle.phase_shift_tk(FK, 1.69e8, kx, ws, geo['dt'], geo['travel_time'], snum, tnum)
    want = np.fft.ifft(TK).real
    assert rel_max(out, want) < F64_TOL, rel_max(out, want)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('profile', ['gradient', 'gradient_with_jumps', 'slowing', 'wavy', 'noisy_steps'])
@pytest.mark.parametrize('snum,tnum', [(700, 48), (1100, 20), (130, 33)])
def test_velocity_that_changes_at_every_step(hip, dtype, profile, snum, tnum, capfd, monkeypatch):
    """A per-step velocity profile handed to the C entry point (what getVelocityProfile returns for anything but a few
    thick layers): no runs of constant velocity, so ps_smooth_kernel carries sqrt(coss) and the rotation from step
    to step (Newton / second-order updates, float64 anchors) and re-does the frequencies near the evanescent boundary
    -- and every frequency at a velocity jump -- from scratch in the reference's rounding.  Against the oracle's
    literal per-step loop (mig_python.py:438-487) at the stated bars; the metrics line names the kernel."""
    import ctypes as C
    import json
    from impdar_amd import _hip, synth
    from oracle import mig_oracle
    lib, ctx = _hip.load(), _hip.context()
    nt = 1 << int(np.ceil(np.log2(snum)))
    geo = synth.geometry(snum, tnum)
    rng = np.random.default_rng(snum + tnum)
    data = (rng.standard_normal((snum, tnum)) + 0.3).astype(dtype)
    u = np.linspace(0., 1., snum)
    vm = {'gradient': 1.69e8 + 0.5e8 * u,
          'gradient_with_jumps': 1.69e8 + 0.3e8 * u + 0.05e8 * (u > 0.3) + 0.08e8 * (u > 0.7),
          'slowing': 2.1e8 - 0.4e8 * u,
          'wavy': 1.8e8 + 0.15e8 * np.sin(7. * u) + 0.1e8 * u,
          'noisy_steps': 1.69e8 + 0.4e8 * u + 2.0e4 * rng.standard_normal(snum)}[profile]
    vm = np.ascontiguousarray(vm, dtype=np.float64)
    kx = mig_oracle._kx(tnum, geo['trace_int'], geo['dist'])
    ws = 2. * np.pi * np.fft.fftfreq(nt, d=geo['dt'])
    out = np.empty((snum, tnum), dtype=dtype)
    tt = np.ascontiguousarray(geo['travel_time'], dtype=np.float64)
    dp = C.POINTER(C.c_double)
    _hip.check(lib.impdar_phaseshift(ctx, data.ctypes.data_as(C.c_void_p), _hip.dtype_code(dtype), snum, tnum, nt,
                                     kx.ctypes.data_as(dp), ws.ctypes.data_as(dp), C.c_double(geo['dt']),
                                     tt.ctypes.data_as(dp), C.c_double(0.0), vm.ctypes.data_as(dp), snum, C.c_double(5.),
                                     C.c_double(7.), out.ctypes.data_as(C.c_void_p)), 'impdar_phaseshift')
    buf = C.create_string_buffer(1024)
    _hip.check(lib.impdar_ctx_last_metrics(ctx, buf, len(buf)), 'metrics')
    assert json.loads(buf.value.decode())['kernel'] == ('ps_smooth32_kernel' if dtype == np.float32 else 'ps_smooth_kernel')
    tap = mig_oracle._apply_taper(data.astype(np.float64), 5, 7, inplace_form=True)
    FK = np.fft.fft2(tap, (nt, tnum))
    TK = mig_oracle.phase_shift_tk(FK, vm, kx, ws, geo['dt'], geo['travel_time'], snum, tnum)
    want = np.fft.ifft(TK).real
    if dtype == np.float32:
        assert rel_l2(out, want) < F32_L2, (profile, rel_l2(out, want))
    else:
        assert rel_max(out, want) < F64_TOL, (profile, rel_max(out, want))


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_transforms_are_repeated_when_the_transform_path_hands_the_call_on(hip, dtype, capfd, monkeypatch):
    """The library lays the spectrum out for ps_nufft_kernel's pairs of wavenumbers (the transform over the traces first, k >= 0
    only) when it expects that kernel to take the call.  When the kernel hands the call on after all, the next kernel reads the
    [k][w > 0] layout: the transforms are made again.  float64: two thick layers with 40 short runs (4 steps each) between them --
    few thick layers, but 160 steps to sum directly, which the kernel declines; float32 (where the library's estimate counts the
    short steps beforehand): the test switch that makes the transform path stand aside.  The same radargram with the thick layers
    alone stays on the transform path, in the half layout.  Against the oracle either way."""
    import ctypes as C
    import json
    from impdar_amd import _hip, synth
    from oracle import mig_oracle
    lib, ctx = _hip.load(), _hip.context()
    snum, tnum = 1024, 128
    nt = 1024
    geo = synth.geometry(snum, tnum)
    rng = np.random.default_rng(5)
    data = (rng.standard_normal((snum, tnum)) + 0.3).astype(dtype)
    kx = mig_oracle._kx(tnum, geo['trace_int'], geo['dist'])
    ws = 2. * np.pi * np.fft.fftfreq(nt, d=geo['dt'])
    tt = np.ascontiguousarray(geo['travel_time'], dtype=np.float64)
    dp = C.POINTER(C.c_double)
    monkeypatch.setenv('IMPDAR_TRACE', '1')
    for handed_on in (True, False):
        short_runs = 40 if (handed_on and dtype == np.float64) else 0
        if handed_on and dtype == np.float32:
            monkeypatch.setenv('IMPDAR_PS_TEST_EDGE_OVERFLOW', '1')
        else:
            monkeypatch.delenv('IMPDAR_PS_TEST_EDGE_OVERFLOW', raising=False)
        vm = np.full(snum, 1.69e8)
        at = 400
        for r in range(short_runs):
            vm[at:at + 4] = 1.70e8 + 0.004e8 * r
            at += 4
        vm[at:] = 1.85e8
        out = np.empty((snum, tnum), dtype=dtype)
        capfd.readouterr()
        _hip.check(lib.impdar_phaseshift(ctx, data.ctypes.data_as(C.c_void_p), _hip.dtype_code(dtype), snum, tnum, nt,
                                         kx.ctypes.data_as(dp), ws.ctypes.data_as(dp), C.c_double(geo['dt']),
                                         tt.ctypes.data_as(dp), C.c_double(0.0), vm.ctypes.data_as(dp), snum, C.c_double(5.),
                                         C.c_double(7.), out.ctypes.data_as(C.c_void_p)), 'impdar_phaseshift')
        err = capfd.readouterr().err
        assert 'the transform over the traces first' in err, err[-2000:]
        assert ('transforms repeated' in err) == handed_on, err[-2000:]
        buf = C.create_string_buffer(1024)
        _hip.check(lib.impdar_ctx_last_metrics(ctx, buf, len(buf)), 'metrics')
        assert (json.loads(buf.value.decode())['kernel'] == 'ps_nufft_kernel') == (not handed_on), buf.value
        tap = mig_oracle._apply_taper(data.astype(np.float64), 5, 7, inplace_form=True)
        FK = np.fft.fft2(tap, (nt, tnum))
        TK = mig_oracle.phase_shift_tk(FK, vm, kx, ws, geo['dt'], geo['travel_time'], snum, tnum)
        want = np.fft.ifft(TK).real
        if dtype == np.float32:
            assert rel_l2(out, want) < F32_L2, (handed_on, rel_l2(out, want))
        else:
            assert rel_max(out, want) < F64_TOL, (handed_on, rel_max(out, want))


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_time_axis_padded_far_beyond_the_samples(hip, dtype):
    """The C entry point takes nt from the caller (any nt >= snum; the package passes the next power of two,
    mig_python.py:262): with nt = 1024 for 100 samples the half spectrum (nt / 2 + 1 rows) is larger than the image,
    which the scratch buffers of the transposed transforms must allow for.  Antisymmetric axes: the half walk runs."""
    import ctypes as C
    from impdar_amd import _hip, synth
    from oracle import mig_oracle
    lib, ctx = _hip.load(), _hip.context()
    snum, tnum, nt = 100, 48, 1024
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=5).astype(dtype)
    kx = mig_oracle._kx(tnum, geo['trace_int'], geo['dist'])
    ws = 2. * np.pi * np.fft.fftfreq(nt, d=geo['dt'])
    out = np.empty((snum, tnum), dtype=dtype)
    tt = np.ascontiguousarray(geo['travel_time'], dtype=np.float64)
    dp = C.POINTER(C.c_double)
    vm = np.where(np.arange(snum) < 40, 1.69e8, 2.0e8).astype(np.float64)
    for vlen in (0, snum):
        _hip.check(lib.impdar_phaseshift(ctx, data.ctypes.data_as(C.c_void_p), _hip.dtype_code(dtype), snum, tnum, nt,
                                         kx.ctypes.data_as(dp), ws.ctypes.data_as(dp), C.c_double(geo['dt']),
                                         tt.ctypes.data_as(dp), C.c_double(1.69e8), vm.ctypes.data_as(dp) if vlen else None, vlen,
                                         C.c_double(5.), C.c_double(7.), out.ctypes.data_as(C.c_void_p)), 'impdar_phaseshift')
        tap = mig_oracle._apply_taper(data.astype(np.float64), 5, 7, inplace_form=True)
        FK = np.fft.fft2(tap, (nt, tnum))
        TK = mig_oracle.phase_shift_tk(FK, vm if vlen else 1.69e8, kx, ws, geo['dt'], geo['travel_time'], snum, tnum)
        want = np.fft.ifft(TK).real
        if dtype == np.float64:
            assert rel_max(out, want) < F64_TOL, rel_max(out, want)
        else:
            assert rel_l2(out, want) < F32_L2, rel_l2(out, want)


def test_velocity_file_and_errors(hip, tmp_path):
    """test/test_migrationlib.py:120-131: constant, layered from a file,
    TypeError for an unreadable file."""
    from impdar_amd.lib.NoInitRadarData import NoInitRadarData
    from impdar_amd.lib import migrationlib
    data = NoInitRadarData(big=True)
    data = migrationlib.migrationPhaseShift(data)
    assert not data.data.any()
    fn = tmp_path / 'velocity_layers.txt'
    fn.write_text('1.677e8  0\n1.677e8  50\n1.2e8  51\n2.2e8  100\n')
    data = NoInitRadarData(big=True)
    data.travel_time = data.travel_time / 10.
    data = migrationlib.migrationPhaseShift(data, vel_fn=str(fn))
    assert data.data.shape == (10, 20)
    data = NoInitRadarData(big=True)
    with pytest.raises(TypeError):
        migrationlib.migrationPhaseShift(data, vel_fn=str(tmp_path / 'notafile.txt'))
    # test_PhaseShiftLateral (test/test_migrationlib.py:133-135): 3-column table, all-zero data
    fn = tmp_path / 'velocity_lateral.txt'
    np.savetxt(fn, golden('P3_velocity_profile')['tab_lat'])
    data = NoInitRadarData(big=True)
    data = migrationlib.migrationPhaseShift(data, vel_fn=str(fn))
    assert data.data.shape == (10, 20) and not data.data.any()
    data = NoInitRadarData(big=True)
    data.data = data.data.astype(int)
    with pytest.raises(TypeError):
        migrationlib.migrationPhaseShift(data)


@pytest.mark.parametrize('snum,tnum,layered', [(200, 96, False), (300, 64, True), (1100, 48, False), (600, 40, True)])
def test_vs_oracle_sizes(hip, snum, tnum, layered):
    """nt = 256 .. 2048: every block-size / frequencies-per-lane instantiation
    up to M = 4."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from oracle import mig_oracle
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=snum)
    if layered:
        Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
        vel = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
    else:
        vel = 1.69e8
    want = mig_oracle.phase_shift(data, geo['dt'], geo['trace_int'], geo['travel_time'], geo['dist'], vel, 20, 30)
    d = RadarData(None)
    d.data, d.snum, d.tnum = data.copy(), snum, tnum
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    from impdar_amd.lib import migrationlib
    migrationlib.migrationPhaseShift(d, vel=vel, htaper=20, vtaper=30)
    assert rel_max(d.data, want) < F64_TOL, rel_max(d.data, want)


@pytest.mark.parametrize('snum,tnum', [(512, 256), (1000, 512)])
def test_float32_boundary_frequencies_in_quiet_tiles(hip, snum, tnum):
    """dx 1 m, dt 10 ns and a first layer at 1.68e8 m/s put (kx, w) pairs exactly on the evanescent boundary
    (coss = 0 to rounding: 2 dx / (v dt) = 25/21; with nt = 2 tnum wavenumber 25 q meets frequency 42 q).  The float32 v(z)
    kernel walks those in fp64 at every step's own velocity, as a correction to the sums of its unrolled tiles
    (mig_python.py:456-485 decides keep-or-drop from the sign of coss step by step); the float64 kernel and the
    oracle evaluate every step anyway.  Both are held to the oracle."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from impdar_amd.lib import migrationlib
    from oracle import mig_oracle
    geo = synth.geometry(snum, tnum)
    assert geo['dt'] == 1e-8 and abs(geo['trace_int'][0] - 1.0) < 1e-12
    data = synth.noise_radargram(snum, tnum, seed=7)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    vel = np.array([[1.68e8, 0.], [1.68e8, 0.45 * Rp], [1.8e8, 0.7 * Rp], [1.9e8, 1.2 * Rp]])
    # the case is what it claims: some (kx, w) of the first layer sit on the boundary
    kx = mig_oracle._kx(tnum, geo['trace_int'], geo['dist'])
    nt = 1 << int(np.ceil(np.log2(snum)))
    ws = 2. * np.pi * np.fft.fftfreq(nt, d=geo['dt'])[1:]
    cs = 1. - (0.5 * 1.68e8 * kx[:, None] / ws[None, :]) ** 2
    assert (np.abs(cs) < 1e-8).sum() >= 4
    want = mig_oracle.phase_shift(data, geo['dt'], geo['trace_int'], geo['travel_time'], geo['dist'], vel, 20, 30)
    for dtype, tol, measure in ((np.float64, F64_TOL, rel_max), (np.float32, F32_L2, rel_l2)):
        d = RadarData(None)
        d.data, d.snum, d.tnum = data.astype(dtype), snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        migrationlib.migrationPhaseShift(d, vel=vel, htaper=20, vtaper=30)
        assert measure(d.data, want) < tol, (dtype, measure(d.data, want))


@pytest.mark.parametrize('snum,tnum', [(1100, 48), (2100, 24)])
def test_float32_vz_four_and_eight_frequencies_per_lane(hip, snum, tnum):
    """float32 layered v(z) at nt = 2048 and 4096: the 4- and 8-frequencies-per-lane instantiations of the
    constant-velocity-runs kernel (the other float32 cases cover 1, 2 and 16)."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from impdar_amd.lib import migrationlib
    from oracle import mig_oracle
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=snum).astype(np.float32)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    vel = np.array([[1.68e8, 0.], [1.68e8, 0.3 * Rp], [1.8e8, 0.6 * Rp], [1.9e8, 1.2 * Rp]])
    want = mig_oracle.phase_shift(data.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'],
                                  geo['dist'], vel, 20, 30)
    d = RadarData(None)
    d.data, d.snum, d.tnum = data.copy(), snum, tnum
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    migrationlib.migrationPhaseShift(d, vel=vel, htaper=20, vtaper=30)
    assert rel_l2(d.data, want) < F32_L2, rel_l2(d.data, want)


def test_float32_velocity_changing_in_most_tiles(hip):
    """A table of 60 thin layers: the velocity moves in more than half of the 16-step tiles, so the host keeps the
    per-step float32 kernel (ps_kernel<float, ..., v(z)>) instead of the constant-velocity-runs kernel."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from impdar_amd.lib import migrationlib
    from oracle import mig_oracle
    snum, tnum = 600, 64
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=11).astype(np.float32)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    nl = 60
    vel = np.stack([np.linspace(1.68e8, 1.9e8, nl), np.linspace(0., 1.2 * Rp, nl)], axis=1)
    vmig = mig_oracle.get_velocity_profile(geo['travel_time'], vel)
    moved = np.abs(np.diff(vmig)) > 1e-10 * np.abs(vmig[1:])
    tiles = np.unique((np.nonzero(moved)[0] + 1) // 16)
    assert 2 * len(tiles) > (snum + 15) // 16           # the case is what it claims
    want = mig_oracle.phase_shift(data.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'],
                                  geo['dist'], vel, 20, 30)
    d = RadarData(None)
    d.data, d.snum, d.tnum = data.copy(), snum, tnum
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    migrationlib.migrationPhaseShift(d, vel=vel, htaper=20, vtaper=30)
    assert rel_l2(d.data, want) < F32_L2, rel_l2(d.data, want)


@pytest.mark.parametrize('snum,tnum', [(2100, 12), (4200, 8)])
def test_float64_vz_eight_and_sixteen_frequencies_per_lane(hip, snum, tnum):
    """float64 layered v(z) at nt = 4096 and 8192: the 8- and 16-frequencies-per-lane instantiations of the float64
    constant-velocity-runs kernel (what a float64 file of that length gets), against the oracle."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from impdar_amd.lib import migrationlib
    from oracle import mig_oracle
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=snum)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    vel = np.array([[1.68e8, 0.], [1.68e8, 0.3 * Rp], [1.8e8, 0.6 * Rp], [1.9e8, 1.2 * Rp]])
    want = mig_oracle.phase_shift(data, geo['dt'], geo['trace_int'], geo['travel_time'], geo['dist'], vel, 20, 30)
    d = RadarData(None)
    d.data, d.snum, d.tnum = data.copy(), snum, tnum
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    migrationlib.migrationPhaseShift(d, vel=vel, htaper=20, vtaper=30)
    assert rel_max(d.data, want) < F64_TOL, rel_max(d.data, want)


@pytest.mark.parametrize('layered', [False, True])
def test_float32_larger_size_vs_oracle(hip, layered):
    """float32 recurrences over ~1000 depth steps (nt = 1024, two frequencies per lane)."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from impdar_amd.lib import migrationlib
    from oracle import mig_oracle
    snum, tnum = 1000, 192
    geo = synth.geometry(snum, tnum)
    data = synth.diffractor_radargram(snum, tnum, ndiff=16).astype(np.float32)
    if layered:
        Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
        vel = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
    else:
        vel = 1.69e8
    want = mig_oracle.phase_shift(data.astype(np.float64), geo['dt'], geo['trace_int'], geo['travel_time'],
                                  geo['dist'], vel, 20, 30)
    d = RadarData(None)
    d.data, d.snum, d.tnum = data.copy(), snum, tnum
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    migrationlib.migrationPhaseShift(d, vel=vel, htaper=20, vtaper=30)
    assert rel_l2(d.data, want) < F32_L2, rel_l2(d.data, want)


def test_tk_is_taper_only(hip):
    g = golden('T1_tk_taper_only')
    dat = make_dat(g)
    dat.migrate('tk', htaper=int(g['htaper']), vtaper=int(g['vtaper']))
    assert dat.flags.mig == 'tk'
    assert np.array_equal(dat.data, g['expected'])
    from impdar_amd.lib.NoInitRadarData import NoInitRadarData
    from impdar_amd.lib import migrationlib
    data = NoInitRadarData(big=True)
    data = migrationlib.migrationTimeWavenumber(data)
    assert not data.data.any()


def _per_step_oracle(data64, geo, vm, htaper=20, vtaper=30):
    """migrationPhaseShift's steps (mig_python.py:253-282) with the per-step velocities given (what getVelocityProfile returns)"""
    from oracle import mig_oracle
    snum, tnum = data64.shape
    tap = mig_oracle._apply_taper(data64, htaper, vtaper, inplace_form=True)
    nt = 2 ** int(np.ceil(np.log2(snum)))
    kx = mig_oracle._kx(tnum, geo['trace_int'], geo['dist'])
    ws = 2. * np.pi * np.fft.fftfreq(nt, d=geo['dt'])
    FK = np.fft.fft2(tap, s=(nt, tnum))
    TK = mig_oracle.phase_shift_tk(FK, vm, kx, ws, geo['dt'], geo['travel_time'], snum, tnum)
    return np.fft.ifft(TK).real, nt, kx, ws


def _run_per_step(hip, data, geo, vm, nt, kx, ws, htaper=20., vtaper=30.):
    import ctypes as C
    import json
    from impdar_amd import _hip
    lib, ctx = _hip.load(), _hip.context()
    snum, tnum = data.shape
    out = np.empty((snum, tnum), dtype=data.dtype)
    tt = np.ascontiguousarray(geo['travel_time'], dtype=np.float64)
    dp = C.POINTER(C.c_double)
    _hip.check(lib.impdar_phaseshift(ctx, data.ctypes.data_as(C.c_void_p), _hip.dtype_code(data.dtype), snum, tnum, nt,
                                     kx.ctypes.data_as(dp), ws.ctypes.data_as(dp), C.c_double(geo['dt']), tt.ctypes.data_as(dp),
                                     C.c_double(0.0), vm.ctypes.data_as(dp), snum, C.c_double(htaper), C.c_double(vtaper),
                                     out.ctypes.data_as(C.c_void_p)), 'impdar_phaseshift')
    buf = C.create_string_buffer(1024)
    _hip.check(lib.impdar_ctx_last_metrics(ctx, buf, len(buf)), 'metrics')
    return out, json.loads(buf.value.decode())['kernel']


def _small_profiles(snum, geo):
    from oracle import mig_oracle
    u = np.linspace(0., 1., snum)
    Rp = 1.9e8 * geo['travel_time'][-1] * 1e-6 / 2.
    tab4 = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
    tab41 = np.stack([np.linspace(1.69e8, 2.2e8, 41), np.linspace(0., 2.0 * Rp, 41)], axis=1)
    return {'gradient': np.ascontiguousarray(1.69e8 + 0.5e8 * u), 'falling': np.ascontiguousarray(2.2e8 - 0.5e8 * u),
            'wavy': np.ascontiguousarray(1.8e8 + 0.15e8 * np.sin(7. * u) + 0.1e8 * u),
            'firn': np.ascontiguousarray(1.69e8 + 0.6e8 * np.exp(-np.arange(snum) * geo['dt'] / 0.8e-6)),
            'jumps': np.ascontiguousarray(1.69e8 + 0.3e8 * u + 0.05e8 * (u > 0.3) + 0.08e8 * (u > 0.7)),
            'vz4': np.ascontiguousarray(mig_oracle.get_velocity_profile(geo['travel_time'], tab4)),
            'layers41': np.ascontiguousarray(mig_oracle.get_velocity_profile(geo['travel_time'], tab41))}


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('snum,tnum', [(300, 64), (520, 40), (1100, 24), (2100, 16), (4200, 8)])
def test_series_path_against_the_oracle(hip, snum, tnum, dtype, monkeypatch):
    """ps_series_kernel (round 6: a velocity that changes inside a piece -- J non-uniform FFTs with shared nodes, direct sums in
    the band above the evanescent boundary) forced onto seven kinds of profile, against the oracle's literal per-step loop
    (mig_python.py:438-487): rising / falling gradients (frequencies dying step by step / none), a wavy profile (both), a firn
    column, jumps inside pieces, smeared layer boundaries of thin and thick layers; record lengths that leave remainders of 8
    and 12 steps (grids of 32 points: the window reaches around them)."""
    from impdar_amd import synth
    monkeypatch.setenv('IMPDAR_PS_MFMA', '7')
    geo = synth.geometry(snum, tnum)
    data64 = synth.noise_radargram(snum, tnum, seed=snum) + 0.5
    for name, vm in _small_profiles(snum, geo).items():
        want, nt, kx, ws = _per_step_oracle(data64, geo, vm)
        got, kernel = _run_per_step(hip, np.ascontiguousarray(data64.astype(dtype)), geo, vm, nt, kx, ws)
        assert kernel == 'ps_series_kernel', (name, kernel)
        if dtype == np.float32:
            assert rel_l2(got, want) < F32_L2, (name, rel_l2(got, want))
            assert rel_l2(got, want) < 2e-6, (name, rel_l2(got, want))       # (observed 2e-7 ... 5e-7)
        else:
            assert rel_max(got, want) < F64_TOL, (name, rel_max(got, want))
            assert rel_max(got, want) < 5e-12, (name, rel_max(got, want))    # (observed 6e-14 ... 3e-12)


@pytest.mark.parametrize('snum,tnum', [(300, 64), (520, 40), (1100, 24), (2100, 16), (4200, 8)])
def test_float64_table_on_the_transform_path_with_its_first_order_term(hip, snum, tnum):
    """A v(z) table on float64 data: ps_nufft_kernel<double> with the runs' velocity noise (2 * gradient(z(t)): ~4e-13, cut at
    1e-11) as the first-order term of the series, in a float32 grid of its own (ps_nufft.h) -- by itself for up to 16 thick
    layers.  Against the oracle at the float64 bar, and against the vector kernels that carried such tables until round 5."""
    from impdar_amd import synth
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=snum + 1) + 0.5
    vm = _small_profiles(snum, geo)['vz4']
    want, nt, kx, ws = _per_step_oracle(data, geo, vm)
    got, kernel = _run_per_step(hip, np.ascontiguousarray(data), geo, vm, nt, kx, ws)
    assert kernel == 'ps_nufft_kernel', kernel
    assert rel_max(got, want) < F64_TOL, rel_max(got, want)
    assert rel_max(got, want) < 5e-12, rel_max(got, want)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('snum,tnum,dt,dx,nl', [(128, 16, 1e-8, 1.0, 7), (200, 64, 5e-9, 2.0, 40), (97, 256, 1.25e-8, 0.5, 70)])
def test_series_path_on_short_records_with_wild_tables(hip, snum, tnum, dt, dx, nl, dtype, monkeypatch):
    """Found by the round-6 fuzz (13 misses of 230 with IMPDAR_PS_MFMA=7): a short record under a table of many layers is ONE piece
    with J = 2, its two grids (8 KB) smaller than the scratch the direct sums keep in the same LDS (12 KB of chunk sums + the
    partial sums) -- which every frequency needs here, the velocity doubling inside the piece.  The host sizes the region for
    both now."""
    from impdar_amd import synth
    from oracle import mig_oracle
    monkeypatch.setenv('IMPDAR_PS_MFMA', '7')
    geo = synth.geometry(snum, tnum, dt=dt, dx=dx)
    rng = np.random.default_rng(nl)
    vs = np.concatenate([[1.68e8], 1.68e8 + np.cumsum(rng.uniform(0.0, 0.06e8, nl - 1))])
    Rv = vs.max() * geo['travel_time'][-1] * 1e-6 / 2.
    zs = np.sort(np.concatenate([[0.], rng.uniform(0.05, 1.2, nl - 1)])) * Rv
    zs[-1] = 1.3 * Rv
    vm = np.ascontiguousarray(mig_oracle.get_velocity_profile(geo['travel_time'], np.stack([vs, zs], axis=1)))
    data64 = synth.noise_radargram(snum, tnum, seed=nl) + 0.5
    want, nt, kx, ws = _per_step_oracle(data64, geo, vm, 7, 9)
    got, kernel = _run_per_step(hip, np.ascontiguousarray(data64.astype(dtype)), geo, vm, nt, kx, ws, 7., 9.)
    assert kernel == 'ps_series_kernel', kernel
    if dtype == np.float32:
        assert rel_l2(got, want) < 5e-6, rel_l2(got, want)
    else:
        assert rel_max(got, want) < F64_TOL, rel_max(got, want)

"""Parity of the HIP vertical_band_pass / constant_space path (csrc/preproc.hip through the C ABI) with the
reference's golden vectors and the CPU oracle, and of the resident chain (data held in HBM from the filter to
the migrated image) with the same steps run through host buffers.

Stated tolerances: band pass, float64 data     max|diff| <= 1e-12 * max|ref| (same fp64 operation order as SciPy)
                   band pass, float32 / int    at most one unit in the last place of the output type
                   constant_space              max|diff| <= 1e-12 * max|ref| (float64 output for every input type)"""
import numpy as np
import pytest

from conftest import golden, golden_names

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _one_transform_implementation(monkeypatch):
    """The first Stolt / phase-shift call of a power-of-two size runs on the library's own row transforms and later calls on
    rocFFT's plans (csrc/own_fft.h): the bit-for-bit host / resident comparisons of this module pin one implementation."""
    monkeypatch.setenv('IMPDAR_PS_FFT', 'own')
    monkeypatch.setenv('IMPDAR_STOLT_FFT', 'own')

TOL = 1e-12


def rel_max(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(float(np.max(np.abs(b))), 1e-300)


def filt_dat(data, dt=1e-8, dist=None):
    from impdar_amd.lib.NoInitRadarData import NoInitRadarDataFiltering
    d = NoInitRadarDataFiltering()
    d.data = data.copy()
    d.snum, d.tnum = data.shape
    d.dt = dt
    d.travel_time = np.arange(d.snum) * dt * 1e6
    d.trace_num = np.arange(d.tnum) + 1.
    for i, attr in enumerate(['lat', 'long', 'x_coord', 'y_coord', 'decday', 'pressure', 'elev']):
        setattr(d, attr, np.arange(d.tnum) * (i + 1.))
    d.trig = np.zeros(d.tnum, dtype=int)
    d.dist = np.arange(d.tnum) / 1000. if dist is None else dist.copy()
    d.trace_int = np.ones(d.tnum)
    return d


def check_filtered(got, want):
    assert got.dtype == want.dtype and got.shape == want.shape
    if got.dtype == np.float64:
        assert rel_max(got, want) < TOL, rel_max(got, want)
    elif got.dtype == np.float32:
        ulp = np.spacing(np.abs(want)).astype(np.float64)
        assert np.all(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= ulp)
    else:
        assert np.max(np.abs(got.astype(np.int64) - want.astype(np.int64))) <= 1
        assert np.mean(got != want) < 1e-3


@pytest.mark.parametrize('name', golden_names('V'))
def test_vertical_band_pass_golden(hip, name):
    g = golden(name)
    d = filt_dat(g['data'], float(g['dt']))
    d.vertical_band_pass(float(g['low']), float(g['high']), order=int(g['order']), filttype=str(g['filttype']),
                         cheb_rp=float(g['cheb_rp']))
    check_filtered(d.data, g['expected'])
    assert np.array_equal(np.asarray(d.flags.bpass, dtype=float), g['bpass'])


@pytest.mark.parametrize('name', golden_names('C'))
def test_constant_space_golden(hip, name):
    g = golden(name)
    d = filt_dat(g['data'], dist=g['dist'])
    for attr in ['lat', 'long', 'x_coord', 'y_coord', 'decday', 'pressure', 'elev', 'trig']:
        setattr(d, attr, g[attr + '_in'].copy())
    d.constant_space(float(g['spacing']), min_movement=float(g['min_movement']))
    assert d.data.dtype == g['expected'].dtype and d.data.shape == g['expected'].shape
    assert rel_max(d.data, g['expected']) < TOL
    assert d.tnum == int(g['tnum_out'])
    for attr in ['lat', 'long', 'x_coord', 'y_coord', 'decday', 'pressure', 'elev', 'trig', 'dist', 'trace_int',
                 'trace_num']:
        assert np.array_equal(getattr(d, attr), g[attr + '_out']), attr
    assert np.array_equal(np.asarray(d.flags.interp, dtype=float), g['interp_flag'])


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('snum,tnum,kw', [
    (700, 131, dict()),
    (1030, 64, dict(filttype='cheb', order=4)),
    (257, 1, dict(filttype='bessel', order=3)),
    (500, 200, dict(order=10)),                    # 21 coefficients
    (400, 67, dict(order=16)),                     # 33 coefficients: the largest the kernel takes
    (300, 300, dict(filttype='fir', order=100)),
    (64, 33, dict(filttype='fir', order=2)),
])
def test_vertical_band_pass_vs_oracle(hip, dtype, snum, tnum, kw):
    from oracle import preproc_oracle as po
    data = np.random.default_rng(snum + tnum).standard_normal((snum, tnum)).astype(dtype)
    d = filt_dat(data)
    d.vertical_band_pass(3., 12., **kw)
    want = po.vertical_band_pass(data, 1e-8, 3., 12., **kw)
    if kw.get('order', 5) >= 10 and kw.get('filttype', 'butter') == 'butter':
        # (b, a) band-pass designs of this order are numerically unstable (SciPy recommends sos): the
        # recurrence amplifies one-ulp differences, so only finiteness-equivalence is asserted
        assert np.array_equal(np.isfinite(d.data), np.isfinite(want))
        good = np.isfinite(want)
        if good.any() and np.max(np.abs(want[good])) < 1e3:
            assert np.max(np.abs(d.data[good].astype(float) - want[good].astype(float))) < 1e-3 * max(1.0, np.max(np.abs(want[good])))
    else:
        check_filtered(d.data, want)


def test_reference_filter_fixtures(hip):
    """test/test_RadarDataFiltering.py:198-229: 500x400 ones through every filter type, the residual bounds
    the reference asserts, and the unknown-type error."""
    from impdar_amd.lib.NoInitRadarData import NoInitRadarDataFiltering
    for ftype, bound in (('butter', 1.0e-1), ('cheb', 1.0e-2), ('bessel', 1.0e-1)):
        d = NoInitRadarDataFiltering()
        d.vertical_band_pass(0.1, 100., filttype=ftype)
        assert np.all(np.abs(d.data) < bound)
    d = NoInitRadarDataFiltering()
    d.vertical_band_pass(1., 10., filttype='fir', order=100)
    d.vertical_band_pass(1., 10., filttype='fir', order=2, fir_window='hanning')
    with pytest.raises(ValueError):
        d.vertical_band_pass(0.1, 100., filttype='dummy')


def test_short_traces_raise_like_scipy(hip):
    d = filt_dat(np.zeros((33, 4)))
    with pytest.raises(ValueError, match='padlen, which is 33'):
        d.vertical_band_pass(3., 12.)


@pytest.mark.parametrize('dtype', [np.float64, np.float32, np.int32, np.complex128])
def test_constant_space_vs_oracle(hip, dtype):
    from oracle import preproc_oracle as po
    rng = np.random.default_rng(7)
    snum, tnum = 300, 1500
    raw = rng.standard_normal((snum, tnum))
    data = (raw * 1e4).astype(dtype) if np.issubdtype(dtype, np.integer) else raw.astype(dtype)
    if dtype == np.complex128:
        data = data + 1.j * rng.standard_normal((snum, tnum))
    steps = 0.5 + rng.random(tnum - 1)
    steps[100:140] = 1e-4
    steps[900] = -0.3                                      # a step backwards counts as "did not move"
    dist = np.hstack(([0.], np.cumsum(steps))) / 1000.
    want, new_dists, _, _ = po.constant_space(data, dist, 0.7)
    d = filt_dat(data, dist=dist)
    d.constant_space(0.7)
    assert d.data.shape == want.shape and d.data.dtype == want.dtype
    assert rel_max(d.data, want) < TOL
    assert np.array_equal(d.dist, new_dists)


def test_reference_constant_space_fixture(hip):
    """test/test_RadarData.py:237-292 on the same kind of object: target size, attribute shapes, a matlab-style
    flags.interp, and fewer traces with a large min_movement."""
    def fresh():
        steps = np.where(np.arange(39) % 3 == 2, 20.0, 35.1)          # metres; every third step is short
        return filt_dat(np.random.default_rng(0).standard_normal((20, 40)), dist=np.hstack(([0.], np.cumsum(steps))) / 1000.)
    d = fresh()
    space = 100.
    targ = int(np.ceil((d.dist[-1] - d.dist[0]) * 1000. / space))
    d.constant_space(space)
    assert d.data.shape == (20, targ)
    for attr in ['x_coord', 'y_coord', 'lat', 'long', 'elev', 'decday']:
        assert getattr(d, attr).shape == (targ,)
    d = fresh()
    d.constant_space(space, min_movement=35.)              # the 20 m steps count as stationary
    assert d.data.shape[0] == 20 and 0 < d.data.shape[1] < targ
    d = fresh()
    d.constant_space(space, min_movement=50.)              # nothing moved: an empty profile, as in the reference
    assert d.data.shape == (20, 0) and d.tnum == 0
    d = fresh()
    d.flags.interp = False
    d.constant_space(space)
    assert d.flags.interp.shape == (2,) and d.flags.interp[0] and d.flags.interp[1] == space


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('mtype', ['stolt', 'kirch', 'phsh', 'tk'])
def test_resident_chain_equals_host_chain(hip, dtype, mtype):
    """band pass -> constant spacing -> migration with the radargram held in HBM gives exactly what the same
    three calls give through host buffers (same kernels, no PCIe round trips in between)."""
    rng = np.random.default_rng(3)
    snum, tnum = 256, 400
    data = rng.standard_normal((snum, tnum)).astype(dtype)
    steps = 0.6 + 0.8 * rng.random(tnum - 1)
    dist = np.hstack(([0.], np.cumsum(steps))) / 1000.

    def chain(resident):
        d = filt_dat(data, dist=dist)
        if resident:
            d.to_device()
            assert d.data is None
        d.vertical_band_pass(2., 20.)
        d.constant_space(1.0)
        d.migrate(mtype, vel=1.69e8, htaper=10, vtaper=10)
        if resident:
            d.from_device()
        return d
    host = chain(False)
    res = chain(True)
    assert res.data.shape == host.data.shape and res.data.dtype == host.data.dtype
    assert res.tnum == host.tnum and res.flags.mig == mtype
    assert np.array_equal(res.data, host.data)
    assert np.isfinite(res.data).all() and np.abs(res.data).max() > 0


def test_resident_band_pass_only_keeps_dtype(hip):
    data = np.random.default_rng(1).standard_normal((200, 50)).astype(np.float32)
    d = filt_dat(data)
    d.to_device()
    d.vertical_band_pass(2., 20.)
    d.from_device()
    h = filt_dat(data)
    h.vertical_band_pass(2., 20.)
    assert d.data.dtype == np.float32 and np.array_equal(d.data, h.data)
    with pytest.raises(TypeError):
        filt_dat(np.zeros((50, 4), dtype=np.int16)).to_device()


def test_full_size_properties(hip):
    """BASELINE config-3 size (4096 x 10000 float32): the band pass is linear and removes a constant, and
    re-spacing a uniformly spaced profile onto its own spacing returns the interior traces unchanged."""
    snum, tnum = 4096, 10000
    rng = np.random.default_rng(0)
    x = rng.standard_normal((snum, tnum)).astype(np.float32)
    y = rng.standard_normal((snum, tnum)).astype(np.float32)

    def vbp(a):
        d = filt_dat(a)
        d.vertical_band_pass(2., 10.)
        return d.data
    fx, fy = vbp(x), vbp(y)
    fz = vbp((x + 2 * y).astype(np.float32))
    err = np.max(np.abs(fz.astype(np.float64) - (fx.astype(np.float64) + 2 * fy.astype(np.float64))))
    assert err < 5e-6 * np.max(np.abs(fz)), err
    const = vbp(np.full((snum, 64), 3.0, dtype=np.float32))
    assert np.max(np.abs(const)) < 1e-3
    d = filt_dat(x, dist=np.arange(tnum) * 1.0e-3)
    d.constant_space(1.0)
    assert d.data.dtype == np.float64 and d.data.shape[0] == snum and abs(d.data.shape[1] - (tnum - 1)) <= 1
    n = d.data.shape[1]
    near = np.abs(d.data - x[:, :n].astype(np.float64))
    assert np.max(near) < 1e-6 * np.max(np.abs(x))


def test_resident_phase_shift_with_velocity_table(hip):
    """migrate('phsh') with a layered (v, z) table on a radargram held in HBM = the host-buffer call."""
    rng = np.random.default_rng(5)
    snum, tnum = 200, 90
    data = rng.standard_normal((snum, tnum)).astype(np.float32)
    tt_end = (snum - 1) * 1e-8
    Rp = 1.9e8 * tt_end / 2.
    tab = np.array([[1.69e8, 0.], [1.69e8, 0.2 * Rp], [1.8e8, 0.5 * Rp], [1.9e8, 1.2 * Rp]])
    import impdar_amd.lib.migrationlib.mig_hip as mh

    def run(resident):
        d = filt_dat(data)
        if resident:
            d.to_device()
            mh._phase_shift(d, tab, None, 10, 10, {}, d._dev)
            d.from_device()
        else:
            mh.migrationPhaseShift(d, vel=tab, htaper=10, vtaper=10)
        return d.data
    a, b = run(False), run(True)
    assert a.dtype == np.float64 and b.dtype == np.float64 and np.array_equal(a, b)


def test_resident_kirchhoff_fast_mode_opt_in(hip, monkeypatch):
    """IMPDAR_KIRCH_MODE=fast on a float64 radargram held in HBM (what constant_space leaves behind): converted
    to float32 on the device and migrated by the LDS-ring kernel, exactly as the host-buffer call with the same
    setting; the result comes back float64."""
    from oracle import c_oracle
    rng = np.random.default_rng(8)
    snum, tnum = 300, 180
    data = rng.standard_normal((snum, tnum))
    monkeypatch.setenv('IMPDAR_KIRCH_MODE', 'fast')
    h = filt_dat(data)
    h.migrate('kirch', vel=1.69e8)
    r = filt_dat(data)
    r.to_device()
    r.migrate('kirch', vel=1.69e8)
    r.from_device()
    assert r.data.dtype == np.float64 and np.array_equal(r.data, h.data)
    want = c_oracle.kirchhoff(data, h.travel_time, h.dist, 1.69e8, False)
    err = np.linalg.norm(r.data - want) / np.linalg.norm(want)
    assert 1e-9 < err < 1e-4, err          # the float32 kernel's error, not the exact kernel's


def test_plan_reused_after_a_device_filter_sees_the_filtered_radargram(hip):
    """Streaming use of the C ABI: a Kirchhoff plan is re-used on a resident radargram
    right after impdar_filtfilt_dev, with no host synchronisation in between.  prep runs on the producer stream
    and must wait for the band pass enqueued on the compute stream (it once read a half-written radargram)."""
    from impdar_amd import _hip, preproc, synth
    from impdar_amd.kirchhoff import KirchhoffPlan
    ctx = hip.context()
    snum, tnum = 2048, 3000                       # the band pass takes long enough to lose a race
    geo = synth.geometry(snum, tnum)
    x = synth.noise_radargram(snum, tnum, seed=8).astype(np.float32)
    spec = preproc.design_filter(geo['dt'], 2., 10.)
    filtered = preproc.filter_host(x.copy(), spec)
    plan = KirchhoffPlan(ctx, np.float32, snum, tnum, geo['dist'], geo['travel_time'], mode='fast')
    d_in = _hip.DeviceArray.from_host(ctx, filtered)
    d_ref = _hip.DeviceArray(ctx, (snum, tnum), np.float32)
    plan.prep(d_in, tnum, 0, tnum)
    plan.migrate(d_ref, 0, tnum)
    plan.sync()
    want = d_ref.to_host()
    for _ in range(3):
        d_x = _hip.DeviceArray.from_host(ctx, x)
        d_out = _hip.DeviceArray(ctx, (snum, tnum), np.float32)
        preproc.filter_dev(d_x, spec)             # enqueued only
        plan.prep(d_x, tnum, 0, tnum)             # same plan, straight after
        plan.migrate(d_out, 0, tnum)
        plan.sync()
        got = d_out.to_host()
        d_x.free()
        d_out.free()
        assert np.array_equal(got, want)
    plan.destroy()
    d_in.free()
    d_ref.free()

"""Parity of the HIP Stolt f-k path (rocFFT + taper/stretch kernels) with the
reference's golden vectors and the CPU oracle.

Stated tolerances: float64 data  max|diff| <= 1e-12 * max|ref|  (observed <= 2.4e-15 over the randomized sweeps,
                   profiles/r04_fuzz.txt; 1e-10 until round 4)
                   float32 data  relative L2 <= 1e-4 (the reference's own
                   float32-vs-float64 difference is ~1e-7)."""
import numpy as np
import pytest

from conftest import golden, golden_names, make_dat, rel_l2, rel_max

pytestmark = pytest.mark.gpu

F64_TOL = 1e-12
F32_L2 = 1e-4


@pytest.mark.parametrize('name', golden_names('S'))
def test_golden(hip, name):
    g = golden(name)
    dat = make_dat(g)
    dat.migrate('stolt', vel=float(g['vel']), htaper=int(g['htaper']), vtaper=int(g['vtaper']))
    assert dat.flags.mig == 'stolt'
    assert dat.data.shape == g['expected'].shape                 # odd snum loses a row (mig_python.py:202)
    assert str(dat.data.dtype) == str(g['expected_dtype'])
    if dat.data.dtype == np.float32:
        assert rel_l2(dat.data, g['expected']) < F32_L2
    else:
        assert rel_max(dat.data, g['expected']) < F64_TOL, rel_max(dat.data, g['expected'])


def test_reference_fixtures(hip):
    """test/test_migrationlib.py:103-110 (zeros, float and int) and
    test/test_impproc.py:603-604 (500x400 ones through the wrapper defaults)."""
    from impdar_amd.lib.NoInitRadarData import NoInitRadarData, NoInitRadarDataFiltering
    from impdar_amd.lib import migrationlib
    from oracle import mig_oracle
    data = NoInitRadarData(big=True)
    data = migrationlib.migrationStolt(data)
    assert data.data.shape == (10, 20) and not data.data.any()
    data = NoInitRadarData(big=True)
    data.data = data.data.astype(int)
    data = migrationlib.migrationStolt(data)
    assert not data.data.any()
    d = NoInitRadarDataFiltering()
    want = mig_oracle.stolt(d.data, d.dt, d.trace_int, None, vel=1.68e8, htaper=100, vtaper=100)
    d.migrate(mtype='stolt', vtaper=100, htaper=100, nxpad=1)
    assert rel_max(d.data, want) < F64_TOL


@pytest.mark.parametrize('snum,tnum,dtype', [(256, 192, np.float64), (1000, 333, np.float64), (513, 64, np.float32),
                                             (2048, 1024, np.float32)])
def test_vs_oracle_sizes(hip, snum, tnum, dtype):
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from oracle import mig_oracle
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=snum).astype(dtype)
    want = mig_oracle.stolt(data, geo['dt'], geo['trace_int'], geo['dist'], 1.68e8, 100, 1000)
    d = RadarData(None)
    d.data, d.snum, d.tnum = data.copy(), snum, tnum
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    d.migrate('stolt', htaper=100, vtaper=1000)
    assert d.data.dtype == want.dtype
    if dtype == np.float32:
        assert rel_l2(d.data, want) < F32_L2
    else:
        assert rel_max(d.data, want) < F64_TOL


@pytest.mark.parametrize('snum,tnum,dtype', [(256, 64, np.float64), (128, 512, np.float64), (1024, 256, np.float32), (64, 2048, np.float32),
                                             (512, 128, np.int16)])
def test_traces_first_on_power_of_two_sizes(hip, snum, tnum, dtype):
    """Power-of-two sizes (64 traces or more) run the transform over the TRACES first, on the radargram's own rows with the taper
    applied on load, keep the wavenumbers k >= 0 with all frequencies and stretch along contiguous rows -- both signs of the
    frequency, the negative half through the mirrored knots of the same row (stolt_stretch_rows; rfft2 / irfft2 at
    mig_python.py:159, 202 keep the frequencies w >= 0 of all wavenumbers instead).  Against the oracle at the stated bars, records
    longer and shorter than they are wide, and int16 data (tapered on the host: the kernel gets no weights)."""
    import ctypes as C
    import json
    from impdar_amd import _hip, synth
    from impdar_amd.lib.RadarData import RadarData
    from oracle import mig_oracle
    geo = synth.geometry(snum, tnum)
    data = synth.noise_radargram(snum, tnum, seed=snum + tnum)
    data = (data * 3000).astype(dtype) if dtype == np.int16 else data.astype(dtype)
    want = mig_oracle.stolt(data, geo['dt'], geo['trace_int'], geo['dist'], 1.68e8, 7, 11)
    d = RadarData(None)
    d.data, d.snum, d.tnum = data.copy(), snum, tnum
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    d.migrate('stolt', htaper=7, vtaper=11)
    buf = C.create_string_buffer(1024)
    _hip.check(_hip.load().impdar_ctx_last_metrics(_hip.context(), buf, len(buf)), 'metrics')
    assert 'stolt_stretch_rows' in json.loads(buf.value.decode())['kernel'], buf.value
    assert d.data.dtype == want.dtype
    if dtype == np.float32:
        assert rel_l2(d.data, want) < F32_L2
    else:
        assert rel_max(d.data, want) < F64_TOL


def test_config2_size_vs_oracle(hip):
    """BASELINE config 2 (4096 x 4096 float32) against the NumPy oracle on the same input."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    from oracle import mig_oracle
    snum = tnum = 4096
    geo = synth.geometry(snum, tnum)
    x = np.random.default_rng(5).standard_normal((snum, tnum)).astype(np.float32)
    want = mig_oracle.stolt(x, geo['dt'], geo['trace_int'], geo['dist'], 1.68e8, 100, 1000)
    d = RadarData(None)
    d.data, d.snum, d.tnum = x.copy(), snum, tnum
    d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
    d.migrate('stolt', htaper=100, vtaper=1000)
    assert d.data.dtype == np.float32 == want.dtype
    assert rel_l2(d.data, want) < 1e-5, rel_l2(d.data, want)


def test_linearity_config2_size(hip):
    """BASELINE config 2 (4096 x 4096 float32): linearity and agreement of
    the float32 path with the float64 path on the same input."""
    from impdar_amd import synth
    from impdar_amd.lib.RadarData import RadarData
    snum = tnum = 4096
    geo = synth.geometry(snum, tnum)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((snum, tnum)).astype(np.float32)
    y = rng.standard_normal((snum, tnum)).astype(np.float32)

    def run(a):
        d = RadarData(None)
        d.data, d.snum, d.tnum = a, snum, tnum
        d.travel_time, d.dist, d.trace_int, d.dt = geo['travel_time'], geo['dist'], geo['trace_int'], geo['dt']
        d.migrate('stolt', htaper=100, vtaper=1000)
        return d.data

    mx, my = run(x), run(y)
    mz = run((2 * x - 3 * y).astype(np.float32))
    assert mx.dtype == np.float32 and np.isfinite(mx).all()
    assert rel_l2(mz, 2.0 * mx.astype(np.float64) - 3.0 * my) < 1e-5
    m64 = run(x.astype(np.float64))
    assert m64.dtype == np.float64
    assert rel_l2(mx, m64) < 1e-5

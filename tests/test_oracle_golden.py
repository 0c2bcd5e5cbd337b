"""The CPU oracle against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from conftest import golden, golden_names, rel_max
from oracle import mig_oracle as o

TOL = 1e-12     # observed <= 5e-16; stated tolerance for the restatement


@pytest.mark.parametrize('name', golden_names('K'))
def test_kirchhoff(name):
    g = golden(name)
    out = o.kirchhoff(g['data'], g['travel_time'], g['dist'], float(g['vel']), bool(g['nearfield']))
    assert out.dtype == np.float64 and out.shape == g['expected'].shape
    assert rel_max(out, g['expected']) < TOL


@pytest.mark.parametrize('name', golden_names('K'))
def test_c_oracle(name):
    """oracle/kirch_oracle.c referees the full-size GPU tests (config 3 / config 4 spot traces) and bench.py's
    in-run parity, so it is pinned to the reference's own outputs too: every Kirchhoff fixture (far and near field,
    uneven spacing, first sample off zero, pre-trigger times, float32 / int16 data, config 1) -- whole image, and
    a subset of output traces (the form those tests use)."""
    from oracle import c_oracle
    g = golden(name)
    near = bool(g['nearfield'])
    out = c_oracle.kirchhoff(g['data'], g['travel_time'], g['dist'], float(g['vel']), near)
    assert out.dtype == np.float64 and out.shape == g['expected'].shape
    assert rel_max(out, g['expected']) < TOL, rel_max(out, g['expected'])
    tr = np.unique(np.linspace(0, g['data'].shape[1] - 1, 5).astype(int))
    sub = c_oracle.kirchhoff(g['data'], g['travel_time'], g['dist'], float(g['vel']), near, traces=tr)
    assert np.array_equal(sub, out[:, tr])
    # the gradient-in form the multi-rank CPU tests drive (a rank's exchanged image)
    tt = np.asarray(g['travel_time']) / 1.0e6
    grad = np.gradient(g['data'], tt, axis=0)
    again = c_oracle.kirchhoff_from_gradient(grad, g['data'], g['travel_time'], g['dist'], float(g['vel']), near)
    assert rel_max(again, g['expected']) < TOL


@pytest.mark.parametrize('name', golden_names('L1'))
def test_kirchhoff_loop_with_the_callers_tables(name):
    """The reference's inner loop called the way its native hook is (mig_cython.h:11): a time limit off max(tt)
    (single-precision rounding, a shortened aperture) and depth tables of the caller's own."""
    g = golden(name)
    out = o.kirchhoff_loop(g['gradD'], g['dist_m'], g['zs'], g['zs2'], g['tt_sec'], float(g['vel']),
                           float(g['max_travel_time']), g['data'], bool(g['nearfield']))
    assert rel_max(out, g['expected']) < TOL


@pytest.mark.parametrize('name', ['K4_kirch_t0_offset', 'K4n_kirch_pretrigger'])
def test_kirchhoff_literal_form(name):
    g = golden(name)
    out = o.kirchhoff_literal(g['data'], g['travel_time'], g['dist'], float(g['vel']), bool(g['nearfield']))
    assert rel_max(out, g['expected']) < TOL


@pytest.mark.parametrize('name', golden_names('S'))
def test_stolt(name):
    g = golden(name)
    out = o.stolt(g['data'], float(g['dt']), g['trace_int'], g['dist'], float(g['vel']),
                  int(g['htaper']), int(g['vtaper']))
    assert out.shape == g['expected'].shape
    assert str(out.dtype) == str(g['expected_dtype'])
    tol = 5e-6 if out.dtype == np.float32 else TOL
    assert rel_max(out, g['expected']) < tol


@pytest.mark.parametrize('name', golden_names('P1') + golden_names('P2') + golden_names('P5'))
def test_phase_shift(name):
    g = golden(name)
    vel = float(g['vel']) if g['vel'].ndim == 0 else g['vel']
    out = o.phase_shift(g['data'], float(g['dt']), g['trace_int'], g['travel_time'], g['dist'], vel,
                        int(g['htaper']), int(g['vtaper']))
    assert rel_max(out, g['expected']) < TOL
    vm = o.get_velocity_profile(g['travel_time'], vel)
    assert np.array_equal(np.asarray(vm), g['vmig'])


@pytest.mark.parametrize('name', golden_names('P4'))
def test_phase_shift_ffd(name):
    # 2-D v(x,z) Fourier finite-difference branch (mig_python.py:428-432,448-487,496-540)
    g = golden(name)
    out = o.phase_shift(g['data'], float(g['dt']), g['trace_int'], g['travel_time'], g['dist'], g['vel'],
                        int(g['htaper']), int(g['vtaper']))
    assert rel_max(out, g['expected']) < TOL
    vm = o.get_velocity_profile(g['travel_time'], g['vel'], g['dist'])
    assert vm.shape == g['vmig'].shape == g['data'].shape
    assert rel_max(vm, g['vmig']) < TOL


def test_velocity_profile():
    g = golden('P3_velocity_profile')
    for c in 'abc':
        vm = o.get_velocity_profile(g['tt_' + c], g['tab_' + c])
        assert np.array_equal(vm, g['vmig_' + c])
    assert o.get_velocity_profile(np.arange(10.), 1.68e8) == 1.68e8
    # 3-column (v, z, x) table (mig_python.py:606-636); cumulative trapezoid instead of the reference's
    # O(snum^2) per-prefix np.trapz: equal to rounding
    for c in ('lat', 'lat2'):
        vm = o.get_velocity_profile(g['tt_' + c], g['tab_lat'], g['dist_' + c])
        assert vm.shape == g['vmig_' + c].shape
        assert np.isfinite(vm).all() and rel_max(vm, g['vmig_' + c]) < TOL
    with pytest.raises(ValueError):            # test/test_migrationlib.py:72-75
        o.get_velocity_profile(g['tt_lat'], g['tab_lat'], None)
    with pytest.raises(ValueError):
        o.get_velocity_profile(g['tt_lat'], g['tab_lat'], np.zeros(20))


def test_velocity_profile_errors():
    tt = np.arange(10.)
    bad = 1.68e8 * np.ones((10, 2))
    bad[:, 1] = 0.
    with pytest.raises(ValueError):
        o.get_velocity_profile(tt, bad)
    for shape in [(8,), (8, 1), (1, 2), (8, 4)]:
        with pytest.raises(ValueError):
            o.get_velocity_profile(tt, 1.68e8 * np.ones(shape))


def test_tk_is_taper_only():
    g = golden('T1_tk_taper_only')
    out = o.time_wavenumber(g['data'], int(g['htaper']), int(g['vtaper']))
    assert np.array_equal(out, g['expected'])


def test_all_zero_fixture_stays_zero():
    # the reference's own migration tests run on 10x20 zeros (test/test_migrationlib.py:103-135)
    z = np.zeros((10, 20))
    tt = np.arange(10.)
    dist = np.arange(20.)
    assert not o.kirchhoff(z, tt, dist).any()
    assert not o.stolt(z, 1, 1, dist).any()
    assert not o.phase_shift(z, 1, 1, tt, dist).any()


def test_shape_check():
    with pytest.raises(ValueError):
        o.check_data_shape(np.ones((1, 1)), 10, 20)
    o.check_data_shape(np.ones((10, 20)), 10, 20)


def test_pair_count_matches_survey():
    # config 1 has 3.2501e7 in-aperture pairs
    assert o.count_pairs(512, 256, 1e-8, 1.0, 1.69e8) == 32500650

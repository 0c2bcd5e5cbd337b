"""RadarData.migrate forwards exactly the keyword arguments the reference's
wrapper forwards (test/test_RadarDataFiltering.py:291-331) and records
flags.mig; runs without a GPU (the migration library is patched)."""
from unittest.mock import patch

import numpy as np
import pytest

from impdar_amd.lib.NoInitRadarData import NoInitRadarData


@patch('impdar_amd.lib.RadarData._RadarDataFiltering.migrationlib')
def test_kwarg_forwarding(lib):
    d = NoInitRadarData(big=True)
    d.migrate(mtype='kirch', vel=1.0, nearfield=True)
    lib.migrationKirchhoff.assert_called_with(d, vel=1.0, nearfield=True)
    assert d.flags.mig == 'kirch'
    d.migrate(mtype='stolt', vel=2.0, htaper=3, vtaper=4)
    lib.migrationStolt.assert_called_with(d, vel=2.0, htaper=3, vtaper=4)
    d.migrate(mtype='phsh', vel=2.0, vel_fn='f', htaper=3, vtaper=4)
    lib.migrationPhaseShift.assert_called_with(d, vel=2.0, vel_fn='f', htaper=3, vtaper=4)
    d.migrate(mtype='tk', vel=2.0, vel_fn='f', htaper=3, vtaper=4)
    lib.migrationTimeWavenumber.assert_called_with(d, vel=2.0, vel_fn='f', htaper=3, vtaper=4)
    d.migrate(mtype='sumigtk', vel=2.0, vel_fn='f', htaper=3, vtaper=4, tmig=5, verbose=6, nxpad=7)
    lib.migrationSeisUnix.assert_called_with(d, mtype='sumigtk', vel=2.0, vel_fn='f', tmig=5, verbose=6, nxpad=7,
                                             htaper=3, vtaper=4)
    assert d.flags.mig == 'sumigtk'


def test_defaults_and_unknown():
    d = NoInitRadarData(big=True)
    with patch('impdar_amd.lib.RadarData._RadarDataFiltering.migrationlib') as lib:
        d.migrate()
        lib.migrationStolt.assert_called_with(d, vel=1.68e8, htaper=10, vtaper=10)
    with pytest.raises(ValueError):
        d.migrate(mtype='bad')
    assert d.flags.mig == 'stolt'


def test_seisunix_missing_binary():
    from impdar_amd.lib import migrationlib
    d = NoInitRadarData(big=True)
    with pytest.raises(FileNotFoundError):
        migrationlib.migrationSeisUnix(d, mtype='sumigtk_not_installed')


def test_gradient_coefficients_match_numpy():
    from impdar_amd.lib.migrationlib.mig_hip import gradient_coefficients
    rng = np.random.default_rng(0)
    f = rng.standard_normal(50)
    for x in (np.arange(50) * 1e-8 * 1e6 / 1e6, np.arange(50.) * 0.5, np.sort(rng.uniform(0, 1, 50))):
        uni, h, ga, gb, gc = gradient_coefficients(x)
        want = np.gradient(f, x)
        got = np.empty(50)
        if uni:
            got[1:-1] = (f[2:] - f[:-2]) / (2. * h)
            got[0] = (f[1] - f[0]) / h
            got[-1] = (f[-1] - f[-2]) / h
        else:
            got[1:-1] = ga[1:-1] * f[:-2] + gb[1:-1] * f[1:-1] + gc[1:-1] * f[2:]
            got[0] = (f[1] - f[0]) / ga[0]
            got[-1] = (f[-1] - f[-2]) / ga[-1]
        assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        gradient_coefficients(np.array([1.0]))


def test_velocity_profile_host_logic():
    """Product getVelocityProfile against the reference's golden outputs and
    its error cases (test/test_migrationlib.py:54-101)."""
    from conftest import golden
    from impdar_amd.lib.migrationlib import getVelocityProfile
    g = golden('P3_velocity_profile')
    d = NoInitRadarData(big=True)
    assert getVelocityProfile(d, 1.68e8) == 1.68e8
    for c in 'abc':
        d = NoInitRadarData(big=True)
        d.travel_time = g['tt_' + c]
        d.snum = len(d.travel_time)
        assert np.array_equal(getVelocityProfile(d, g['tab_' + c]), g['vmig_' + c])
    d = NoInitRadarData(big=True)
    bad = 1.68e8 * np.ones((10, 2))
    bad[:, 1] = 0.
    with pytest.raises(ValueError):
        getVelocityProfile(d, bad)
    # 3-column (v, z, x) table: the reference's fixture on its own test geometry and on a physical one
    # (test/test_migrationlib.py:69-75)
    for c in ('lat', 'lat2'):
        d = NoInitRadarData(big=True)
        d.travel_time = g['tt_' + c]
        d.dist = g['dist_' + c]
        d.snum, d.tnum = len(d.travel_time), len(d.dist)
        vm = getVelocityProfile(d, g['tab_lat'])
        assert vm.shape == (d.snum, d.tnum)
        assert np.allclose(vm, g['vmig_' + c], rtol=1e-12, atol=0)
    d = NoInitRadarData(big=True)
    d.dist = None
    with pytest.raises(ValueError):
        getVelocityProfile(d, g['tab_lat'])
    d.dist = None
    with pytest.raises(ValueError):
        getVelocityProfile(d, 1.68e8 * np.ones((10, 3)))
    d = NoInitRadarData(big=True)
    for shape in [(8,), (8, 1), (1, 2), (8, 4)]:
        with pytest.raises(ValueError):
            getVelocityProfile(d, 1.68e8 * np.ones(shape))

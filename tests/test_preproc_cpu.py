"""vertical_band_pass / constant_space without a GPU: the oracle against the golden vectors
produced by the reference itself (tests/golden/make_golden.py), and the host-side logic of the product
(filter design, stationary-shot bookkeeping, gather tables, per-trace attribute interpolation)."""
import numpy as np
import pytest

from conftest import golden, golden_names
from oracle import preproc_oracle as po
from impdar_amd import preproc

def rel_max(a, b):
    """max |a - b| / max |b|, complex-safe."""
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) / max(float(np.max(np.abs(b))), 1e-300)


TOL = 1e-12     # observed: bit-identical for the band pass, <= 2e-16 for the interpolation


def _kw(g):
    return dict(order=int(g['order']), filttype=str(g['filttype']), cheb_rp=float(g['cheb_rp']))


@pytest.mark.parametrize('name', golden_names('V'))
def test_oracle_vertical_band_pass(name):
    g = golden(name)
    out = po.vertical_band_pass(g['data'], float(g['dt']), float(g['low']), float(g['high']), **_kw(g))
    assert out.dtype == g['expected'].dtype and out.shape == g['expected'].shape
    assert rel_max(out.astype(float), g['expected'].astype(float)) < TOL


@pytest.mark.parametrize('name', golden_names('C'))
def test_oracle_constant_space(name):
    g = golden(name)
    new, new_dists, good_vals, _ = po.constant_space(g['data'], g['dist'], float(g['spacing']), float(g['min_movement']))
    assert new.shape == g['expected'].shape and new.dtype == g['expected'].dtype
    assert rel_max(new, g['expected']) < TOL
    assert np.array_equal(new_dists, g['dist_out'])


def test_oracle_filtfilt_short_trace_raises():
    from scipy import signal
    b, a = signal.butter(5, [0.04, 0.2], 'bandpass')
    with pytest.raises(ValueError, match='padlen, which is 33'):
        po.filtfilt(b, a, np.zeros((33, 2)))


@pytest.mark.parametrize('name', golden_names('V'))
def test_design_matches_oracle(name):
    g = golden(name)
    spec = preproc.design_filter(float(g['dt']), float(g['low']), float(g['high']), **_kw(g))
    kind, b, a = po.design(float(g['dt']), float(g['low']), float(g['high']), **_kw(g))
    assert spec[0] == kind
    assert np.array_equal(spec[1], b)
    if kind == 'iir':
        assert np.array_equal(spec[2], a)
        assert np.array_equal(spec[3], po.lfilter_zi(b, a))


def test_design_rejects_unknown_filter():
    with pytest.raises(ValueError, match='not recognized'):
        preproc.design_filter(1e-8, 1., 10., filttype='dummy')


@pytest.mark.parametrize('name', golden_names('C'))
def test_spacing_plan_and_attributes(name):
    """Everything constant_space does on the host: good_vals, corrected distances, new distances, the gather
    tables handed to the device (checked by evaluating them with NumPy) and the per-trace attributes."""
    g = golden(name)
    dist = g['dist'].copy()
    plan = preproc.SpacingPlan(dist, float(g['spacing']), float(g['min_movement']))
    assert np.array_equal(plan.new_dists, g['dist_out'])
    assert plan.n_new == int(g['tnum_out'])
    data = g['data']
    y = data if np.issubdtype(data.dtype, np.inexact) else data.astype(np.float64)
    manual = (y[:, plan.hi] - y[:, plan.lo]) / plan.den[None, :] * plan.t[None, :] + y[:, plan.lo]
    assert rel_max(manual, g['expected']) < TOL
    for attr in ['lat', 'long', 'x_coord', 'y_coord', 'decday', 'pressure', 'elev', 'trig']:
        got = preproc.interp1d_linear(plan.temp_dist, g[attr + '_in'][plan.good_vals], plan.new_dists)
        assert np.array_equal(got, g[attr + '_out']), attr
    trace_int = np.hstack((np.array(np.nanmean(np.diff(plan.new_dists))), np.diff(plan.new_dists))) * 1000.
    assert np.array_equal(trace_int, g['trace_int_out'])


def test_interp1d_linear_range_errors_and_slope_form():
    x = np.array([0., 1., 3.])
    with pytest.raises(ValueError, match='below the interpolation range'):
        preproc.interp1d_linear(x, x, np.array([-0.1]))
    with pytest.raises(ValueError, match='above the interpolation range'):
        preproc.interp1d_linear(x, x, np.array([3.5]))
    y32 = np.array([1., 2., 6.], dtype=np.float32)        # float32 values take SciPy's slope form
    got = preproc.interp1d_linear(x, y32, np.array([0.5, 2.0]))
    assert got.dtype == np.float64 and np.allclose(got, [1.5, 4.0])
    # unsorted abscissae are sorted first (stable), as interp1d does
    got = preproc.interp1d_linear(np.array([3., 0., 1.]), np.array([6., 1., 2.]), np.array([2.0]))
    assert np.allclose(got, [4.0])


def test_spacing_plan_without_movement_is_empty():
    """All shots stationary: the reference's arange(min, max) is empty and so is the re-spaced profile."""
    plan = preproc.SpacingPlan(np.array([0., 0., 0.]), 1.0)
    assert plan.n_new == 0 and plan.good_vals.tolist() == [True, False, False]

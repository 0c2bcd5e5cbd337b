import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    """Load tests/golden/<name>.npz as a dict."""
    path = os.path.join(GOLDEN, name + '.npz')
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


class Dat(object):
    """Bare attribute bag with the RadarData fields the migrations touch."""


def make_dat(g):
    from impdar_amd.lib.RadarData import RadarData
    d = RadarData(None)
    d.data = g['data'].copy()
    d.snum, d.tnum = d.data.shape
    d.travel_time = g['travel_time'].copy()
    d.dist = g['dist'].copy()
    d.trace_int = g['trace_int'].copy()
    d.dt = float(g['dt'])
    d.fn = ''
    return d


def rel_max(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b)) / max(np.max(np.abs(b)), 1e-300))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.fixture(scope='session')
def hip():
    """The loaded HIP library bindings; GPU tests fail (not skip) without it."""
    from impdar_amd import _hip
    _hip.load()
    assert _hip.device_count() > 0, 'no GPU visible'
    return _hip

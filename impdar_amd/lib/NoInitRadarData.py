"""Synthetic RadarData objects for tests, with the shapes and attribute values of the test doubles in the
reference's ``src/impdar/lib/NoInitRadarData.py:30-90`` (so its migration and filtering tests can be restated
one-for-one), built from small tables instead of a ``.mat`` file."""
import numpy as np

from .RadarData import RadarData
from .RadarFlags import RadarFlags


def _fill(obj, data, scalars, per_trace):
    """``data`` plus the bookkeeping every RadarData needs: ``per_trace`` maps an attribute to a function of
    the trace index vector, ``scalars`` to plain values."""
    obj.data = data
    obj.snum, obj.tnum = data.shape
    obj.fn = ''
    obj.chan = 1
    obj.trig_level = 0.
    idx = np.arange(obj.tnum)
    for name, make in per_trace.items():
        setattr(obj, name, make(idx))
    for name, value in scalars.items():
        setattr(obj, name, value)
    return obj


class NoInitRadarData(RadarData):
    """A 2 x 2 radargram of small integers, or with ``big=True`` 10 samples x 20 traces of zeros; unit trace
    spacing and sample interval."""

    def __init__(self, big=False):
        super(NoInitRadarData, self).__init__(None)
        if big:
            data = np.zeros((10, 20))
            tt = np.arange(10)
        else:
            data = np.array([[2, 2], [1, 1]])
            tt = 0.001 * np.arange(2) + 0.001
        _fill(self, data, dict(trace_int=1, dt=1, travel_time=tt),
              dict(dist=lambda i: i, elevation=lambda i: np.zeros(len(i)), long=lambda i: i * 3., lat=lambda i: i * 2.,
                   trace_num=lambda i: i + 1., decday=lambda i: i.astype(float), trig=lambda i: np.zeros(len(i)),
                   pressure=lambda i: np.zeros(len(i))))


class NoInitRadarDataFiltering(RadarData):
    """500 samples x 400 traces of ones sampled every nanosecond (the reference's filtering fixture)."""

    def __init__(self):
        super(NoInitRadarDataFiltering, self).__init__(None)
        dt = 0.001e-6
        _fill(self, np.ones((500, 400)), dict(dt=dt, travel_time=0.001 * np.arange(500) + 0.001, flags=RadarFlags()),
              dict(trace_num=lambda i: i + 1., trace_int=lambda i: dt * np.ones(len(i)), long=lambda i: i * 3.,
                   lat=lambda i: i * 2., x_coord=lambda i: i * 3., y_coord=lambda i: i * 2., decday=lambda i: i,
                   elev=lambda i: i * 0.001 + 100, trig=lambda i: np.zeros(len(i)).astype(int),
                   pressure=lambda i: np.zeros(len(i))))

"""Synthetic RadarData test doubles with the same shapes and attribute values
as the reference's ``src/impdar/lib/NoInitRadarData.py:30-90`` so its tests
can be restated one-for-one."""
import numpy as np

from .RadarData import RadarData
from .RadarFlags import RadarFlags


class NoInitRadarData(RadarData):
    """2x2 by default; ``big=True`` gives 10 samples x 20 traces of zeros."""

    def __init__(self, big=False):
        super(NoInitRadarData, self).__init__(None)
        if big:
            self.data = np.zeros((10, 20))
            self.travel_time = np.arange(self.data.shape[0])
        else:
            self.data = np.array([[2, 2], [1, 1]])
            self.travel_time = 0.001 * np.arange(self.data.shape[0]) + 0.001
        self.fn = ''
        self.snum, self.tnum = self.data.shape
        self.dist = np.arange(self.tnum)
        self.elevation = np.zeros((self.tnum,))
        self.long = np.arange(self.tnum) * 3.
        self.lat = np.arange(self.tnum) * 2.
        self.trace_num = np.arange(self.tnum) + 1.
        self.decday = np.arange(self.tnum).astype(float)
        self.trace_int = 1
        self.dt = 1
        self.trig = np.zeros((self.tnum,))
        self.pressure = np.zeros((self.tnum,))
        self.chan = 1
        self.trig_level = 0.


class NoInitRadarDataFiltering(RadarData):
    """500 samples x 400 traces of ones (the reference's filtering fixture)."""

    def __init__(self):
        super(NoInitRadarDataFiltering, self).__init__(None)
        self.fn = ''
        self.data = np.ones((500, 400))
        self.snum, self.tnum = self.data.shape
        self.travel_time = 0.001 * np.arange(self.snum) + 0.001
        self.trace_num = np.arange(self.tnum) + 1.
        self.dt = 0.001e-6
        self.trace_int = self.dt * np.ones((self.tnum,))
        self.flags = RadarFlags()
        self.long = np.arange(self.tnum) * 3.
        self.lat = np.arange(self.tnum) * 2.
        self.x_coord = np.arange(self.tnum) * 3.
        self.y_coord = np.arange(self.tnum) * 2.
        self.decday = np.arange(self.tnum)
        self.elev = np.arange(self.tnum) * 0.001 + 100
        self.trig = np.zeros_like(self.elev).astype(int)
        self.pressure = np.zeros((self.tnum,))
        self.chan = 1
        self.trig_level = 0.

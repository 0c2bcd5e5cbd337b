"""Processing-history flags carried by a RadarData object.

Mirrors the attribute contract of the reference's
``src/impdar/lib/RadarFlags.py:44-98`` (names, defaults, ``.mat`` struct
layout) because ``RadarData.migrate`` records ``flags.mig`` and ``save()``
serialises the whole struct.
"""
import numpy as np


class RadarFlags(object):
    #: (name, preallocated length or None) in the order the .mat struct stores them
    _SPEC = (('batch', None), ('bpass', 3), ('hfilt', 2), ('rgain', None), ('agc', None),
             ('restack', None), ('reverse', None), ('crop', 3), ('nmo', 2), ('interp', 2),
             ('mig', None), ('elev', None))
    bool_attrs = ['agc', 'batch', 'restack', 'reverse', 'rgain']

    def __init__(self):
        self.batch = False
        self.bpass = np.zeros((3,))
        self.hfilt = np.zeros((2,))
        self.rgain = False
        self.agc = False
        self.restack = False
        self.reverse = False
        self.crop = np.zeros((3,))
        self.nmo = np.zeros((2,))
        self.interp = np.zeros((2,))
        self.mig = 'none'
        self.elev = 0
        self.elevation = 0
        self.attrs = [name for name, _ in self._SPEC]
        self.attr_dims = [dim for _, dim in self._SPEC] + [None, None]

    def to_matlab(self):
        """dict for :func:`scipy.io.savemat` (booleans become 0/1)."""
        out = {}
        for name in self.attrs:
            val = getattr(self, name)
            if name in self.bool_attrs:
                val = 1 if val else 0
            out[name] = val
        return out

    def from_matlab(self, matlab_struct):
        """Fill from the ``flags`` struct of :func:`scipy.io.loadmat` output."""
        for name, dim in self._SPEC:
            val = matlab_struct[name][0][0][0]
            if dim is not None and np.shape(val)[0] == 1:
                val = np.zeros((dim,))
            setattr(self, name, val)
        for name in self.bool_attrs:
            setattr(self, name, True if matlab_struct[name][0][0][0] == 1 else 0)

    def __repr__(self):
        return 'RadarFlags(' + ', '.join('%s=%r' % (a, getattr(self, a)) for a in self.attrs) + ')'

"""Minimal ``RadarData`` container: the attribute contract the migration path
reads and writes, the ``.mat`` (StoDeep/ImpDAR) round trip and the
``migrate`` dispatch.

Mirrors (re-implemented, not copied) the reference's
``src/impdar/lib/RadarData/__init__.py:36-61`` (attribute lists), ``:124-244``
(None-initialisation and ``.mat`` load), ``:270-331`` (``check_attrs``),
``_RadarDataSaving.py:32-78`` (``save`` incl. the cast back to the file's
dtype), ``_RadarDataFiltering.py:590-637`` (``migrate``) and the two steps an
impproc chain runs in front of a migration: ``vertical_band_pass``
(``_RadarDataFiltering.py:469-549``) and ``constant_space``
(``_RadarDataProcessing.py:499-583``).  Everything else in the reference's
class (other filters, picks, GPS, plotting) is out of scope.
"""
import numpy as np

from ..ImpdarError import ImpdarError
from ..RadarFlags import RadarFlags
from ._RadarDataFiltering import migrate as _migrate, vertical_band_pass as _vertical_band_pass
from ._RadarDataProcessing import constant_space as _constant_space
from ... import resident as _resident

STODEEP_ATTRS = ['data', 'migdata', 'interp_data', 'nmo_data', 'filtdata', 'hfilt_data']


class RadarData(object):
    attrs_guaranteed = ['chan', 'data', 'decday', 'dt', 'pressure', 'snum', 'tnum', 'trace_int',
                        'trace_num', 'travel_time', 'trig', 'trig_level']
    attrs_optional = ['nmo_depth', 'lat', 'long', 'elev', 'dist', 'x_coord', 'y_coord', 'fn', 't_srs']
    stodeep_attrs = STODEEP_ATTRS

    migrate = _migrate
    vertical_band_pass = _vertical_band_pass
    constant_space = _constant_space
    to_device = _resident.to_device
    from_device = _resident.from_device

    def __init__(self, fn_mat):
        for attr in self.attrs_guaranteed + self.attrs_optional:
            setattr(self, attr, None)
        self.flags = RadarFlags()
        self.picks = None
        self.data_dtype = None
        self._picks_struct = None
        self._dev = None
        self.fn = fn_mat
        if fn_mat is None:
            return
        from scipy.io import loadmat
        mat = loadmat(fn_mat)
        for attr in self.attrs_guaranteed:
            if attr == 'data':
                self._take_data(mat)
            elif attr not in mat:
                raise KeyError('.mat file does not appear to be in the StoDeep/ImpDAR format')
            else:
                setattr(self, attr, self._unbox(mat[attr], strict2d=True))
        for attr in self.attrs_optional:
            setattr(self, attr, self._unbox(mat[attr], strict2d=False) if attr in mat else None)
        self.data_dtype = self.data.dtype
        self.fn = fn_mat
        self.flags = RadarFlags()
        self.flags.from_matlab(mat['flags'])
        if 'picks' in mat:
            self._picks_struct = mat['picks']     # carried through save() untouched
        self.check_attrs()

    @staticmethod
    def _unbox(val, strict2d):
        """loadmat returns everything 2-D: scalars -> python scalars, vectors -> 1-D."""
        if val.shape == (1, 1):
            return val[0][0]
        if val.shape[0] == 1 or (len(val.shape) > 1 and val.shape[1] == 1):
            return val.flatten()
        return val

    def _take_data(self, mat):
        """First available of the StoDeep data matrices becomes ``data``."""
        for i, name in enumerate(self.stodeep_attrs):
            if name in mat:
                val = mat[name]
                if len(val.dtype) > 0:
                    print('Warning: Multiple arrays stored in {:s}, taking the first.'.format(name))
                    val = val[0][0][0]
                if i > 0:
                    print('First priority data {:s} not in structure, using {:s}'.format(
                        self.stodeep_attrs[0], name))
                self.data = val
                return
        raise KeyError('Data do not appear to be in StoDeep format')

    def check_attrs(self):
        """Raise ImpdarError for an ill-defined object."""
        for attr in self.attrs_guaranteed + ['fn']:
            if not hasattr(self, attr):
                raise ImpdarError('{:s} is missing. It appears that this is an ill-defined RadarData object'.format(attr))
            if getattr(self, attr) is None:
                raise ImpdarError('{:s} is None. It appears that this is an ill-defined RadarData object'.format(attr))
        for attr in self.attrs_optional:
            if not hasattr(self, attr):
                raise ImpdarError('{:s} is missing. It appears that this is an ill-defined RadarData object'.format(attr))
        if (self.data.shape != (self.snum, self.tnum)) and (self.elev is None):
            raise ImpdarError('The data shape does not match the snum and tnum values!!!')
        for attr in ['lat', 'long', 'pressure', 'trig', 'elev', 'dist', 'x_coord', 'y_coord', 'decday']:
            val = getattr(self, attr, None)
            if val is None:
                continue
            if (not hasattr(val, 'shape')) or len(val.shape) < 1:
                if val == 0:
                    setattr(self, attr, None)       # matlab's stand-in for None
                elif attr == 'trig':
                    self.trig = np.ones((self.tnum,), dtype=int) * int(self.trig)
                else:
                    raise ImpdarError('{:s} needs to be a vector'.format(attr))
            elif val.shape[0] != self.tnum:
                raise ImpdarError('{:s} needs length tnum {:d}'.format(attr, self.tnum))
        if getattr(self, 'data_dtype', None) is None:
            self.data_dtype = self.data.dtype

    def save(self, fn):
        """Write a StoDeep/ImpDAR ``.mat``; ``data`` is cast back to the dtype
        it was loaded with (NaN-aware for integer files)."""
        from scipy.io import savemat
        if getattr(self, '_dev', None) is not None:
            self.from_device()
        mat = {}
        for attr in self.attrs_guaranteed:
            val = getattr(self, attr)
            mat[attr] = val if val is not None else 0
        for attr in self.attrs_optional + self.stodeep_attrs:
            if getattr(self, attr, None) is not None:
                mat[attr] = getattr(self, attr)
        if self._picks_struct is not None:
            mat['picks'] = self._picks_struct
        mat['flags'] = (self.flags if self.flags is not None else RadarFlags()).to_matlab()
        want = getattr(self, 'data_dtype', None)
        if want is not None and want != mat['data'].dtype:
            has_nan = np.issubdtype(mat['data'].dtype, np.floating) and np.any(np.isnan(mat['data']))
            if want in [int, np.int8, np.int16] and has_nan:
                print('Warning: new file is float16 rather than ', want, ' since we now have NaNs')
                mat['data'] = mat['data'].astype(np.float16)
            elif want in [np.int32] and has_nan:
                print('Warning: new file is float32 rather than ', want, ' since we now have NaNs')
                mat['data'] = mat['data'].astype(np.float32)
            elif want in [np.int64] and has_nan:
                print('Warning: new file is float64 rather than ', want, ' since we now have NaNs')
                mat['data'] = mat['data'].astype(np.float64)
            else:
                mat['data'] = mat['data'].astype(want)
        savemat(fn, mat)

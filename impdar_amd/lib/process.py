"""The band-pass / re-spacing / migration chain of ImpDAR's multi-step driver.

Mirrors the part of the reference's ``src/impdar/lib/process.py`` that leads up to a migration, in the
reference's order (``:72-197``): ``vbp=(low, high)`` -> ``dat.vertical_band_pass`` (``:151-154``),
``interp=(spacing, gps_fn)`` -> ``gpslib.interp`` = ``dat.constant_space`` when no GPS file is given
(``:178-180``), ``migrate=X`` -> ``dat.migrate(mtype='stolt')`` whatever ``X`` is (``:190-193`` -- the string
given on the command line is ignored by the reference, and so it is here).  When more than one of the steps
is requested on a float radargram it is uploaded once and stays in HBM until the last step is done.
``process_and_exit`` loads, processes and saves with the reference's file naming (``:30-70``, ``:274-295``).
The other steps of that driver (crop, nmo, horizontal filters, restack, denoise, reverse) are out of scope
and rejected.
"""
import os

import numpy as np

from .load import load

_OUT_OF_SCOPE = ('rev', 'hfilt', 'ahfilt', 'nmo', 'crop', 'hcrop', 'restack', 'denoise')


def process(RadarDataList, interp=None, vbp=None, migrate=None, **kwargs):
    """Returns True if something was done (reference ``process.py:72-197``)."""
    for name in _OUT_OF_SCOPE:
        if kwargs.get(name) not in (None, False):
            raise NotImplementedError('processing step %r is not part of the MI355X migration engine; '
                                      'run it with the reference ImpDAR first' % name)
    if vbp is not None:
        if not hasattr(vbp, '__iter__'):
            raise TypeError('vbp must be a tuple with first two elements [low] [high] MHz')
    if interp is not None:
        try:
            float(interp[0])
            interp[1]
        except (ValueError, TypeError, IndexError):
            raise ValueError('interp must be a target spacing (float) then a gps filename')
        if interp[1] is not None:
            raise NotImplementedError('kinematic GPS control (a gps filename for interp) is not part of the '
                                      'MI355X migration engine')
    steps = sum(x is not None for x in (vbp, interp, migrate))
    if steps == 0:
        return False
    for dat in RadarDataList:
        resident = steps > 1 and np.asarray(dat.data).dtype in (np.float32, np.float64)
        if resident:
            dat.to_device()
        try:
            if vbp is not None:
                dat.vertical_band_pass(*vbp)
            if interp is not None:
                dat.constant_space(float(interp[0]))
            if migrate is not None:
                dat.migrate(mtype='stolt')
        finally:
            if resident:
                dat.from_device()
    return True


def _save(rd_list, outpath=None, cat=False):
    if outpath is not None:
        if len(rd_list) > 1:
            for rd in rd_list:
                bn = os.path.split(os.path.splitext(rd.fn)[0])[1]
                if bn[-4:] == '_raw':
                    bn = bn[:-4]
                rd.save(os.path.join(outpath, bn + '_proc.mat'))
        else:
            rd_list[0].save(outpath)
    else:
        for rd in rd_list:
            bn = os.path.splitext(rd.fn)[0]
            if bn[-4:] == '_raw':
                bn = bn[:-4]
            rd.save(bn + ('.mat' if cat else '_proc.mat'))


def process_and_exit(fn, cat=False, filetype='mat', o=None, **kwargs):
    if cat:
        raise NotImplementedError('concatenation is not part of the MI355X migration engine')
    radar_data = load(filetype, fn)
    processed = process(radar_data, **kwargs)
    if not processed:
        print('No processing steps performed. Not saving!')
    else:
        _save(radar_data, outpath=o, cat=cat)

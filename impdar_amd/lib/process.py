"""The migrate hook of ImpDAR's multi-step driver.

Mirrors the part of the reference's ``src/impdar/lib/process.py`` that touches
migration: ``process(..., migrate=X)`` calls ``dat.migrate(mtype='stolt')`` for
every RadarData whatever ``X`` is (``process.py:190-193`` -- the string given
on the command line is ignored by the reference, and so it is here),
``process_and_exit`` loads, processes and saves with the reference's file
naming (``:30-70``, ``:274-295``).  The other processing steps of that driver
(crop, nmo, filters, restack, interp, denoise) are out of scope and rejected.
"""
import os

from .load import load

_OUT_OF_SCOPE = ('interp', 'rev', 'vbp', 'hfilt', 'ahfilt', 'nmo', 'crop', 'hcrop', 'restack', 'denoise')


def process(RadarDataList, migrate=None, **kwargs):
    """Returns True if something was done (reference ``process.py:72-197``)."""
    for name in _OUT_OF_SCOPE:
        if kwargs.get(name) not in (None, False):
            raise NotImplementedError('processing step %r is not part of the MI355X migration engine; '
                                      'run it with the reference ImpDAR first' % name)
    done_stuff = False
    if migrate is not None:
        for dat in RadarDataList:
            dat.migrate(mtype='stolt')
        done_stuff = True
    return done_stuff


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

"""Loader for the one file type on the migration CLI path: ImpDAR/StoDeep
``.mat`` (reference ``src/impdar/lib/load/__init__.py:28,79-80``).  The
reference's other 14 instrument readers are out of scope; convert with the
reference's ``impdar load`` first."""
from .RadarData import RadarData

FILETYPE_OPTIONS = ['mat']


def load(filetype, fns_in, channel=1, *args, **kwargs):
    if not isinstance(fns_in, (list, tuple)):
        fns_in = [fns_in]
    if filetype != 'mat':
        raise ValueError('Only ImpDAR .mat files can be loaded here (got %s); '
                         'convert other formats with the reference\'s `impdar load`' % filetype)
    return [RadarData(fn) for fn in fns_in]

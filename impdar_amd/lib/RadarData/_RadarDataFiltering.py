"""``RadarData.migrate``: the string dispatch into the migration library
(reference ``src/impdar/lib/RadarData/_RadarDataFiltering.py:590-637``: same
mtype names, per-mtype keyword forwarding, defaults, ValueError for unknown
names, ``flags.mig`` recorded afterwards)."""
from .. import migrationlib


def migrate(self, mtype='stolt', vtaper=10, htaper=10, tmig=0, vel_fn=None, vel=1.68e8,
            nxpad=10, nearfield=False, verbose=0):
    """Migrate the data in place.  mtype: 'kirch', 'stolt', 'phsh', 'tk' or 'su*'."""
    if mtype == 'kirch':
        migrationlib.migrationKirchhoff(self, vel=vel, nearfield=nearfield)
    elif mtype == 'stolt':
        migrationlib.migrationStolt(self, vel=vel, htaper=htaper, vtaper=vtaper)
    elif mtype == 'phsh':
        migrationlib.migrationPhaseShift(self, vel=vel, vel_fn=vel_fn, htaper=htaper, vtaper=vtaper)
    elif mtype == 'tk':
        migrationlib.migrationTimeWavenumber(self, vel=vel, vel_fn=vel_fn, htaper=htaper, vtaper=vtaper)
    elif mtype[:2] == 'su':
        migrationlib.migrationSeisUnix(self, mtype=mtype, vel=vel, vel_fn=vel_fn, tmig=tmig,
                                       verbose=verbose, nxpad=nxpad, htaper=htaper, vtaper=vtaper)
    else:
        raise ValueError('Unrecognized migration routine')
    self.flags.mig = mtype

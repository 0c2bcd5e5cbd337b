"""``RadarData.migrate``: the string dispatch into the migration library
(reference ``src/impdar/lib/RadarData/_RadarDataFiltering.py:590-637``: same
mtype names, per-mtype keyword forwarding, defaults, ValueError for unknown
names, ``flags.mig`` recorded afterwards), and ``RadarData.vertical_band_pass``
(``:469-549``), the filter an impproc chain runs in front of a migration."""
from .. import migrationlib
from ... import preproc


def vertical_band_pass(self, low, high, order=5, filttype='butter', cheb_rp=5, fir_window='hamming',
                       *args, **kwargs):
    """Band-pass every trace along time between ``low`` and ``high`` MHz: forward-backward IIR
    (butter / cheb / bessel) or a delayed-and-shifted FIR; same designs, padding and initial conditions as
    the reference (SciPy), the filtering itself on the MI355X."""
    spec = preproc.design_filter(self.dt, low, high, order=order, filttype=filttype, cheb_rp=cheb_rp)
    print('Bandpassing from {:4.1f} to {:4.1f} MHz...'.format(low, high))
    dev = getattr(self, '_dev', None)
    if dev is not None:
        preproc.filter_dev(dev, spec)
    else:
        self.data = preproc.filter_host(self.data, spec)
    print('Bandpass filter complete.')
    self.flags.bpass[0] = 1
    self.flags.bpass[1] = low
    self.flags.bpass[2] = high


def migrate(self, mtype='stolt', vtaper=10, htaper=10, tmig=0, vel_fn=None, vel=1.68e8,
            nxpad=10, nearfield=False, verbose=0):
    """Migrate the data in place.  mtype: 'kirch', 'stolt', 'phsh', 'tk' or 'su*'."""
    if getattr(self, '_dev', None) is not None:
        # radargram held in HBM (to_device): Kirchhoff and Stolt run on it where it is
        from ... import resident
        if mtype == 'kirch':
            resident.kirchhoff_resident(self, vel=vel, nearfield=nearfield)
        elif mtype == 'stolt':
            resident.stolt_resident(self, vel=vel, htaper=htaper, vtaper=vtaper)
        elif mtype == 'phsh':
            migrationlib.mig_hip._phase_shift(self, vel, vel_fn, htaper, vtaper, {}, self._dev)
        else:
            self.from_device()
            migrate(self, mtype=mtype, vtaper=vtaper, htaper=htaper, tmig=tmig, vel_fn=vel_fn, vel=vel,
                    nxpad=nxpad, nearfield=nearfield, verbose=verbose)
            self.to_device()
            return
        self.flags.mig = mtype
        return
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

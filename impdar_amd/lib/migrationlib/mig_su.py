"""SeisUnix migrations are out of scope (they shell out to third-party
binaries, reference ``src/impdar/lib/migrationlib/mig_su.py:32-171``).  The
entry point is kept so ``mtype='su*'`` fails the way the reference fails on a
machine without SeisUnix (``mig_su.py:80-81``)."""
import shutil


def migrationSeisUnix(dat, mtype='sumigtk', vel=1.69e8, vel_fn=None, tmig=0, verbose=1, nxpad=100,
                      htaper=100, vtaper=1000, nz=None, dz=None, quiet=False):
    if shutil.which(mtype) is None:
        raise FileNotFoundError('Cannot find chosen SeisUnix migration routine,' + mtype +
                                '. Either install or choose a different migration routine.')
    raise NotImplementedError('SeisUnix-backed migration (%s) is not part of the MI355X engine; '
                              'use kirch, stolt or phsh' % mtype)

#! /usr/bin/env python
"""``impproc migrate`` (and the two steps usually run in front of it, ``vbp`` and ``interp``) on the MI355X
engine.

Mirrors these sub-commands of the reference's ``src/impdar/bin/impproc.py`` (migrate parser ``:295-343``,
vbp ``:113-125``, interp ``:222-251``, ``main`` ``:378-415``, ``mig`` ``:508-519``, ``vbp`` ``:438-440``,
``interp`` ``:483-491``): same options, types and defaults, same output naming
(``<name minus _raw>_<migrated|bandpassed|interp>.mat``, ``-o`` file or folder).  The reference's other
processing sub-commands are out of scope.

    python -m impdar_amd.bin.impproc migrate --mtype kirch line1_raw.mat
"""
import argparse
import os
import sys

from ..lib.load import load, FILETYPE_OPTIONS


def _get_args():
    parser = argparse.ArgumentParser()
    subparsers = parser.add_subparsers(help='Choose a processing step')
    parser_mig = subparsers.add_parser('migrate', help='Migration')
    parser_mig.set_defaults(func=mig, name='migrated')
    parser_mig.add_argument('--mtype', type=str, default='phsh',
                            choices=['stolt', 'kirch', 'phsh', 'tk', 'sumigtk', 'sustolt', 'sumigffd'],
                            help='Migration routines.')
    parser_mig.add_argument('--vel', type=float, default=1.69e8,
                            help='Speed of light in dielectric medium m/s (default is for ice, 1.69e8)')
    parser_mig.add_argument('--vel_fn', type=str, default=None,
                            help='Filename for input velocity array. Column 1: velocities, '
                                 'Column 2: z locations, Column 3: x locations (optional)')
    parser_mig.add_argument('--nearfield', action='store_true',
                            help='Boolean for nearfield operator in Kirchhoff migration.')
    parser_mig.add_argument('--htaper', type=int, default=100, help='Number of samples for horizontal taper')
    parser_mig.add_argument('--vtaper', type=int, default=1000, help='Number of samples for vertical taper')
    parser_mig.add_argument('--nxpad', type=int, default=100, help='Number of traces to pad with zeros for FFT')
    parser_mig.add_argument('--tmig', type=int, default=0, help='Times for velocity profile')
    parser_mig.add_argument('--verbose', type=int, default=1, help='Print output from SeisUnix migration')
    parser_mig.add_argument('--gpus', type=int, default=0,
                            help='(extension) shard a Kirchhoff (output-trace blocks) or constant-v / v(z) phase-shift (wavenumber '
                                 'slabs) migration over this many MI355X of the node '
                                 '(default: $IMPDAR_NGPUS, else one)')
    _add_def_args(parser_mig)

    parser_vbp = subparsers.add_parser('vbp', help='Vertically bandpass the data')
    parser_vbp.set_defaults(func=vbp, name='bandpassed')
    parser_vbp.add_argument('low_MHz', type=float, help='Lowest frequency passed (in MHz)')
    parser_vbp.add_argument('high_MHz', type=float, help='Highest frequency passed (in MHz)')
    _add_def_args(parser_vbp)

    parser_interp = subparsers.add_parser('interp', help='Reinterpolate GPS')
    parser_interp.set_defaults(func=interp, name='interp')
    parser_interp.add_argument('spacing', type=float, help='New spacing of radar traces, in meters')
    parser_interp.add_argument('--gps_fn', type=str, default=None,
                               help='File with precision GPS (kinematic GPS control is not part of this engine; '
                                    'only the default, the GPS already in the file, is accepted)')
    parser_interp.add_argument('--offset', type=float, default=0.0, help='Offset from GPS time to radar time')
    parser_interp.add_argument('--minmove', type=float, default=1.0e-2, help='Minimum movement to not be stationary')
    parser_interp.add_argument('--extrapolate', action='store_true', help='Extrapolate GPS data beyond bounds')
    _add_def_args(parser_interp)
    return parser


def _add_def_args(parser):
    parser.add_argument('fns', type=str, nargs='+', help='The files to process')
    parser.add_argument('-o', type=str, help='Output to this file (folder if multiple inputs)')
    parser.add_argument('--ftype', type=str, default='mat', help='Type of file to load (default ImpDAR mat)',
                        choices=FILETYPE_OPTIONS)


def main():
    parser = _get_args()
    args = parser.parse_args(sys.argv[1:])
    if not hasattr(args, 'func'):
        parser.parse_args(['-h'])

    radar_data = load(args.ftype, args.fns)
    if args.name == 'interp':
        interp(radar_data, **vars(args))
    else:
        for dat in radar_data:
            args.func(dat, **vars(args))

    if args.o is not None:
        if (len(radar_data) > 1) or (args.o[-1] == '/'):
            for d, f in zip(radar_data, args.fns):
                bn = os.path.split(os.path.splitext(f)[0])[1]
                if bn[-4:] == '_raw':
                    bn = bn[:-4]
                d.save(os.path.join(args.o, bn + '_{:s}.mat'.format(args.name)))
        else:
            radar_data[0].save(args.o)
    else:
        for d, f in zip(radar_data, args.fns):
            bn = os.path.splitext(f)[0]
            if bn[-4:] == '_raw':
                bn = bn[:-4]
            d.save(bn + '_{:s}.mat'.format(args.name))


def mig(dat, mtype='stolt', vel=1.69e8, vtaper=100, htaper=100, tmig=0, verbose=0, vel_fn=None, nxpad=1,
        nearfield=False, gpus=0, **kwargs):
    """Migrate data (defaults as the reference's ``impproc.mig``)."""
    if gpus and gpus > 1:
        os.environ['IMPDAR_NGPUS'] = str(gpus)
    dat.migrate(mtype, vel=vel, vtaper=vtaper, htaper=htaper, tmig=tmig, verbose=verbose, vel_fn=vel_fn,
                nxpad=nxpad, nearfield=nearfield)


def vbp(dat, low_MHz=1, high_MHz=10000, **kwargs):
    """Vertically bandpass the data."""
    dat.vertical_band_pass(low_MHz, high_MHz)


def interp(dats, spacing, gps_fn=None, offset=0.0, minmove=1.0e-2, extrapolate=False, **kwargs):
    """Move data to constant spacing (the reference's ``gpslib.interp`` without external GPS control)."""
    if gps_fn is not None:
        raise NotImplementedError('kinematic GPS control (--gps_fn) is not part of the MI355X migration engine')
    for dat in dats:
        dat.constant_space(spacing, min_movement=minmove)


if __name__ == '__main__':
    main()

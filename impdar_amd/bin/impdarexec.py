#! /usr/bin/env python
"""``impdar proc [-vbp LOW HIGH] [-interp SPACING GPS_FN] [-migrate X] files`` on the MI355X engine (reference
``src/impdar/bin/impdarexec.py:47-119,175-182`` -> ``process.process_and_exit``).  Only the ``proc``
sub-command with these three steps is provided."""
import argparse
import sys

from ..lib import process


def _get_args():
    parser = argparse.ArgumentParser()
    subparsers = parser.add_subparsers(help='Choose a processing step')
    parser_proc = subparsers.add_parser('proc', help='Process data')
    parser_proc.set_defaults(func=process.process_and_exit)
    parser_proc.add_argument('-vbp', nargs=2, type=float,
                             help='Bandpass the data vertically at low (MHz) and high (MHz)')
    parser_proc.add_argument('-interp', nargs=2, type=str,
                             help='Reinterpolate GPS. First argument is the new spacing, in meters. Second argument '
                                  'is the filename with new GPS data (not supported by this engine)')
    parser_proc.add_argument('-migrate', type=str, help='Migrate with the indicated routine.')
    parser_proc.add_argument('fn', type=str, nargs='+', help='File(s) to process')
    parser_proc.add_argument('-o', type=str, help='Write to this filename')
    return parser


def main():
    parser = _get_args()
    args = parser.parse_args(sys.argv[1:])
    if not hasattr(args, 'func'):
        parser.parse_args(['-h'])
        return None
    kw = vars(args)
    func = kw.pop('func')
    return func(**kw)


if __name__ == '__main__':
    main()

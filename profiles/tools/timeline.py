#!/usr/bin/env python3
"""Timeline of the LAST burst of GPU activity in a rocprofv3 --kernel-trace --memory-copy-trace output directory
(a burst = events separated from the previous ones by more than 20 ms): every kernel and copy with start / end in ms
from the burst's first event.   usage: timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv
import glob
import os
import sys


def main():
    ev = []
    for f in glob.glob(os.path.join(sys.argv[1], '**', '*_kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
    for f in glob.glob(os.path.join(sys.argv[1], '**', '*_memory_copy_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY %s %s bytes' % (r.get('Direction', ''), r.get('Bytes', r.get('Size', '?')))))
    ev.sort()
    start = 0
    for i in range(1, len(ev)):
        if ev[i][0] - max(e for _, e, _ in ev[max(0, i - 50):i]) > 20e6:
            start = i
    burst = ev[start:]
    t0 = burst[0][0]
    # merge runs of equal consecutive names (the pieces of a staged copy)
    out = []
    for s, e, k in burst:
        if out and out[-1][2] == k and s - out[-1][1] < 0.2e6:
            out[-1] = (out[-1][0], max(e, out[-1][1]), k, out[-1][3] + 1)
        else:
            out.append((s, e, k, 1))
    for s, e, k, n in out:
        print('%8.3f .. %8.3f  (%6.3f ms)  %s%s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, k, ' x%d' % n if n > 1 else ''))


if __name__ == '__main__':
    main()

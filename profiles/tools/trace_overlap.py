#!/usr/bin/env python3
"""Read a rocprofv3 --kernel-trace CSV and say, for every RCCL kernel, when it ran relative to the diffraction-sum
kernels: start / end against the kirch_quad_kernel launch in flight at that moment (if any) and the share of its own
duration that lies inside diffraction-sum kernels.   usage: trace_overlap.py <kernel_trace.csv> [max lines]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
    ev.sort()
    t0 = ev[0][0]
    quad = [(s, e) for s, e, k in ev if 'kirch_quad_kernel' in k or 'kirch_dquad_kernel' in k]
    rccl = [(s, e, k) for s, e, k in ev if 'nccl' in k.lower() or 'rccl' in k.lower()]
    print('%d kernels, %d diffraction-sum launches, %d RCCL kernels; times in ms from the first kernel' % (len(ev), len(quad), len(rccl)))
    shown = 0
    inside_total = dur_total = 0
    for s, e, k in rccl:
        inside = sum(max(0, min(e, qe) - max(s, qs)) for qs, qe in quad)
        inside_total += inside
        dur_total += e - s
        cur = [(qs, qe) for qs, qe in quad if qs <= s < qe]
        if shown < limit:
            where = ('starts %.3f ms into a diffraction sum that runs %.3f .. %.3f' % ((s - cur[0][0]) / 1e6, (cur[0][0] - t0) / 1e6, (cur[0][1] - t0) / 1e6)
                     if cur else 'starts with no diffraction sum in flight')
            print('%-40s %.3f .. %.3f (%.3f ms), %3.0f %% inside diffraction-sum kernels; %s'
                  % (k[:40], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, 100.0 * inside / max(e - s, 1), where))
            shown += 1
    if dur_total:
        print('all RCCL kernels: %.3f ms in total, %.0f %% of it underneath diffraction-sum kernels' % (dur_total / 1e6, 100.0 * inside_total / dur_total))
    for qs, qe in quad[:limit]:
        print('diffraction sum %.3f .. %.3f (%.3f ms)' % ((qs - t0) / 1e6, (qe - t0) / 1e6, (qe - qs) / 1e6))


if __name__ == '__main__':
    main()

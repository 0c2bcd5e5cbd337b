#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/<run>/{stats,pmc_fetch,pmc_write,pmc_sq})
into the small CSV/JSON files committed next to this script.

    python profiles/summarize.py gpurun_out/prof12 r01

FETCH_SIZE calibration: MI355X_MICROARCH.md (HBM section) says gfx950 under-reports read bytes
(exactly 1/2 for 16 B/lane streams) and that other access widths must be calibrated on a known
byte count in the same access pattern.  kirch_prep_kernel reads the (snum, tnum) float32 input
exactly once with the same dword-per-lane coalesced loads the migration kernel's staging uses,
so known_bytes / FETCH_SIZE(prep) is used as the read correction for the migration kernel.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    files = sorted(glob.glob(os.path.join(d, '*', '*_counter_collection.csv')), key=os.path.getmtime)
    for f in files[-1:]:                       # the merged scratch dir may hold older runs too
        for r in csv.DictReader(open(f)):
            out[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
            meta[r['Kernel_Name']] = dict(vgpr=r['VGPR_Count'], sgpr=r['SGPR_Count'], lds=r['LDS_Block_Size'],
                                          scratch=r['Scratch_Size'], wg=r['Workgroup_Size'], grid=r['Grid_Size'])
    return out, meta


def main():
    src, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    snum, tnum = 4096, 10000
    stats = sorted(glob.glob(os.path.join(src, 'stats', '*', '*_kernel_stats.csv')), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(here, '%s_bench_kernel_stats.csv' % tag))
    rows = []
    allc = {}
    for sub in ('pmc_fetch', 'pmc_write', 'pmc_sq'):
        c, meta = counters(os.path.join(src, sub))
        for k, cs in c.items():
            for name, v in cs.items():
                rows.append((name, k, len(v), sum(v) / len(v), meta[k]))
                allc.setdefault(k, {})[name] = sum(v) / len(v)
    with open(os.path.join(here, '%s_bench_pmc.csv' % tag), 'w') as fo:
        fo.write('counter,kernel,dispatches,mean_value,vgpr,sgpr,lds_block,scratch,workgroup,grid\n')
        for name, k, n, v, m in sorted(rows):
            fo.write('"%s","%s",%d,%r,%s,%s,%s,%s,%s,%s\n' % (name, k, n, v, m['vgpr'], m['sgpr'], m['lds'],
                                                               m['scratch'], m['wg'], m['grid']))
    mig = [k for k in allc if 'kirch_quad_kernel' in k or 'kirch_tab_kernel' in k][0]
    prep = [k for k in allc if 'kirch_prep' in k][0]
    known = snum * tnum * 4
    cal = known / (allc[prep]['FETCH_SIZE'] * 1024)
    traffic = (allc[mig]['FETCH_SIZE'] * cal + allc[mig]['WRITE_SIZE']) * 1024
    out = dict(kernel=mig, fetch_size_kb_raw=allc[mig]['FETCH_SIZE'], write_size_kb_raw=allc[mig]['WRITE_SIZE'],
               fetch_calibration=cal,
               calibration_note='kirch_prep_kernel FETCH_SIZE %.1f KB for a known %.1f KB read (same dword-per-lane '
                                'coalesced loads); see profiles/summarize.py' % (allc[prep]['FETCH_SIZE'], known / 1024),
               hbm_bytes_per_launch=traffic,
               hbm_bytes_per_launch_if_x2=(allc[mig]['FETCH_SIZE'] * 2 + allc[mig]['WRITE_SIZE']) * 1024,
               sq={k: v for k, v in allc[mig].items() if k.startswith('SQ_')},
               source='profiles/%s_bench_pmc.csv (rocprofv3 --pmc, separate passes per counter group, '
                      'bench.py --steps 3 --warmup 1 --no-cpu)' % tag)
    json.dump(out, open(os.path.join(here, 'kirch_fast_hbm_traffic.json'), 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

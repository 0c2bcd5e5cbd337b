#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/<run>/{stats,pmc_fetch,pmc_write,pmc_sq,pmc_sq2},
written by profiles/tools/profile_bench.sh) into the small CSV/JSON files committed next to this script.

    python profiles/summarize.py gpurun_out/prof_r01b r01

FETCH_SIZE correction: MI355X_MICROARCH.md (HBM section) says gfx950 reports exactly 1/2 of the bytes
of wide coalesced 16 B/lane reads and that other access widths must be calibrated on a known byte
count in the same pattern.  The migration kernel's memory reads are all 16 B/lane (LDS-DMA staging of
the image, pick-table rows), so its FETCH_SIZE is doubled.  kirch_prep_kernel reads the (snum, tnum)
float32 input exactly once with dword-per-lane loads; its known byte count is kept in the output as a
cross-check of the counter (ratio known/counted, 1.545 on this part).
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
    for sub in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_sq2'):
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
    traffic = (allc[mig]['FETCH_SIZE'] * 2 + allc[mig]['WRITE_SIZE']) * 1024
    out = dict(kernel=mig, fetch_size_kb_raw=allc[mig]['FETCH_SIZE'], write_size_kb_raw=allc[mig]['WRITE_SIZE'],
               fetch_correction=2.0,
               correction_note='all reads of the kernel are 16 B/lane (LDS-DMA staging, pick rows): gfx950 FETCH_SIZE '
                               'counts half of those bytes (MI355X_MICROARCH.md, HBM section)',
               prep_cross_check='kirch_prep_kernel (4 B/lane reads): FETCH_SIZE %.1f KB for a known %.1f KB read, '
                                'ratio %.3f' % (allc[prep]['FETCH_SIZE'], known / 1024, cal),
               hbm_bytes_per_launch=traffic,
               other={k: v for k, v in allc[mig].items() if not k.endswith('_SIZE')},
               source='profiles/%s_bench_pmc.csv (rocprofv3 --pmc, separate passes per counter group, '
                      'bench.py --steps 3 --warmup 1 --no-cpu; profiles/tools/profile_bench.sh)' % tag)
    json.dump(out, open(os.path.join(here, 'kirch_fast_hbm_traffic.json'), 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Condense the rocprofv3 output of profiles/tools/profile_bench.sh (gpurun_out/<run>/{stats,stats64,paths,pmc_*,paths_sq*})
into the small CSV / JSON files committed next to this script.

    python profiles/summarize.py gpurun_out/prof_r02 r02

FETCH_SIZE correction: MI355X_MICROARCH.md (HBM section) says gfx950 reports exactly 1/2 of the bytes of wide coalesced
16 B/lane reads and that other access widths must be calibrated on a known byte count in the same pattern.  The float32
migration kernel's memory reads are all 16 B/lane (LDS-DMA staging of the image, pick-table rows), so its FETCH_SIZE is
doubled (bench.py does the same for the figure it measures in-run).
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
    for f in glob.glob(os.path.join(d, 'run', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
            meta[r['Kernel_Name']] = dict(vgpr=r['VGPR_Count'], sgpr=r['SGPR_Count'], lds=r['LDS_Block_Size'],
                                          scratch=r['Scratch_Size'], wg=r['Workgroup_Size'], grid=r['Grid_Size'])
    return out, meta


def first(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    for sub, name in (('stats', 'bench'), ('stats64', 'bench_f64'), ('paths', 'paths')):
        f = first(os.path.join(src, sub, 'run', '**', '*kernel_stats.csv'))
        if f:
            shutil.copy(f, os.path.join(here, '%s_%s_kernel_stats.csv' % (tag, name)))
    if os.path.exists(os.path.join(src, 'bench_under_rocprof.json')):
        shutil.copy(os.path.join(src, 'bench_under_rocprof.json'), os.path.join(here, '%s_bench_n1.json' % tag))
    # per-launch durations of the headline kernel in launch order: the stats average above also holds the warm-up
    # launches and the one-shot (end_to_end) launches of a cold plan; the timed region is launches warmup+1 .. warmup+steps
    tr = first(os.path.join(src, 'stats', 'run', '**', '*kernel_trace.csv'))
    bj = os.path.join(src, 'bench_under_rocprof.json')
    if tr and os.path.exists(bj):
        b = json.loads(open(bj).read().strip().splitlines()[-1])
        kname = b['roofline']['kernel']
        ls = sorted((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
                    for r in csv.DictReader(open(tr)) if kname in r['Kernel_Name'])
        d = [x[1] for x in ls]
        w, k = int(b['warmup']), int(b['steps'])
        timed = d[w:w + k]
        with open(os.path.join(here, '%s_bench_kernel_launches.txt' % tag), 'w') as fo:
            fo.write('%s: %d launches in this rocprofv3 --kernel-trace run of `python3 bench.py --no-pmc`, ms each, launch order\n'
                     % (kname, len(d)))
            fo.write(' '.join('%.3f' % x for x in d) + '\n')
            fo.write('average of all launches (what --stats reports): %.4f ms\n' % (sum(d) / len(d)))
            if len(timed) == k:
                fo.write('average of the timed region (launches %d..%d): %.4f ms; bench.py reported kernel_ms = %.4f from HIP '
                         'events in the same process\n' % (w + 1, w + k, sum(timed) / k, b['roofline']['kernel_ms']))
    rows, allc = [], {}
    for sub, fn in (('pmc_fetch', 'bench'), ('pmc_write', 'bench'), ('pmc_sq', 'bench'), ('pmc_sq2', 'bench'),
                    ('paths_sq', 'paths'), ('paths_sq2', 'paths'), ('paths_fetch', 'paths'), ('paths_write', 'paths')):
        c, meta = counters(os.path.join(src, sub))
        for k, cs in c.items():
            for name, v in cs.items():
                rows.append((fn, name, k, len(v), sum(v) / len(v), meta[k]))
                allc.setdefault(k, {})[name] = sum(v) / len(v)
    for fn in ('bench', 'paths'):
        with open(os.path.join(here, '%s_%s_pmc.csv' % (tag, fn)), 'w') as fo:
            fo.write('counter,kernel,dispatches,mean_value,vgpr,sgpr,lds_block,scratch,workgroup,grid\n')
            for f, name, k, n, v, m in sorted(rows):
                if f == fn and any(s in k for s in ('kirch', 'ps_', 'stolt', 'fft', 'transpose')):
                    fo.write('"%s","%s",%d,%r,%s,%s,%s,%s,%s,%s\n' % (name, k, n, v, m['vgpr'], m['sgpr'], m['lds'],
                                                                       m['scratch'], m['wg'], m['grid']))
    mig = [k for k in allc if 'kirch_quad_kernel' in k]
    if mig and 'FETCH_SIZE' in allc[mig[0]] and 'WRITE_SIZE' in allc[mig[0]]:
        a = allc[mig[0]]
        out = dict(kernel=mig[0], fetch_size_kb_raw=a['FETCH_SIZE'], write_size_kb_raw=a['WRITE_SIZE'], fetch_correction=2.0,
                   hbm_bytes_per_launch=(a['FETCH_SIZE'] * 2 + a['WRITE_SIZE']) * 1024,
                   other={k: v for k, v in a.items() if not k.endswith('_SIZE')},
                   source='profiles/%s_bench_pmc.csv (rocprofv3 --pmc, separate passes; profiles/tools/profile_bench.sh)' % tag)
        if 'GRBM_GUI_ACTIVE' in a and 'SQ_LDS_IDX_ACTIVE' in a:
            out['lds_busy_share_of_cu_cycles'] = a['SQ_LDS_IDX_ACTIVE'] / 256.0 / (a['GRBM_GUI_ACTIVE'] / 8.0)
        json.dump(out, open(os.path.join(here, 'kirch_fast_hbm_traffic.json'), 'w'), indent=1)
        print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

#!/bin/bash
# GPU box: durations and SQ counters of the ps_smooth kernels (linear gradient at 8192^2, float32 and float64).  usage: ps_smooth_pmc.sh <outdir under gpurun_out>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
B="python3 $R/profiles/tools/ps_smooth.py 8192"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats/run -o x --output-format csv -- $B > $O/stats.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/sq1/run -o x --output-format csv -- $B > $O/sq1.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS SQ_WAVES SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_IFETCH -d $O/sq2/run -o x --output-format csv -- $B > $O/sq2.log 2>&1 </dev/null
python3 - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
for f in glob.glob(o + '/stats/run/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'ps_smooth' in r['Name']: print('stats', r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
for d in ('sq1', 'sq2'):
    for f in glob.glob(o + '/' + d + '/run/*counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:40]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
        for k in acc:
            if 'ps_smooth' in k and 'sum' not in k:
                print(d, k, 'dispatches', len(n[k]), {c: '%.4g' % (v / len(n[k])) for c, v in acc[k].items()})
PY

#!/bin/bash
# GPU box: SQ counters of the config-5 phase shift's frequency-sum kernel.  usage: ps_pair_pmc.sh <outdir under gpurun_out>  (IMPDAR_PS_MFMA picks the kernel)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
B="python3 $R/profiles/tools/ps_quick.py 8192 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats/run -o x --output-format csv -- $B > $O/stats.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/sq1/run -o x --output-format csv -- $B > $O/sq1.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY -d $O/sq2/run -o x --output-format csv -- $B > $O/sq2.log 2>&1 </dev/null
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_TRANS SQ_WAVES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_COEXEC_CYCLES -d $O/sq3/run -o x --output-format csv -- $B > $O/sq3.log 2>&1 </dev/null
python3 - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
for f in glob.glob(o + '/stats/run/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'ps_' in r['Name']: print('stats', r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
for d in ('sq1', 'sq2', 'sq3'):
    for f in glob.glob(o + '/' + d + '/run/*counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:30]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
        for k in acc:
            if 'ps_pair' in k or 'ps_mfma' in k:
                print(d, k, 'dispatches', len(n[k]), {c: '%.4g' % (v / len(n[k])) for c, v in acc[k].items()})
PY

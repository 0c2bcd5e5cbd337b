#!/bin/bash
# counters of ps_smooth_kernel at 4096^2 (GPU box): profiles/tools/gen_pmc.sh <outdir under gpurun_out>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
B="python3 $R/profiles/tools/ps_smooth.py 4096"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/sq1/run -o x --output-format csv -- $B > $O/sq1.log 2>&1 </dev/null
timeout 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d $O/sq2/run -o x --output-format csv -- $B > $O/sq2.log 2>&1 </dev/null
timeout 200 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS SQ_WAVES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY -d $O/sq3/run -o x --output-format csv -- $B > $O/sq3.log 2>&1 </dev/null
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
for d in ('sq1', 'sq2', 'sq3'):
    for f in glob.glob(o + '/' + d + '/run/*counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:40]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k in acc:
            print(d, k, {c: '%.4g' % v for c, v in acc[k].items()})
PY

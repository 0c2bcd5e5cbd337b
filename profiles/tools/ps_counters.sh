#!/bin/bash
# Run on the GPU box: durations and SQ counters of the config-5 phase-shift kernels (constant v and v(z)).
#   profiles/tools/ps_counters.sh <outdir under gpurun_out>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/stats/run -o x --output-format csv -- python3 profiles/tools/ps_quick.py 8192 2 > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/sq/run -o x --output-format csv -- python3 profiles/tools/ps_quick.py 8192 1 > $O/sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $O/sq2/run -o x --output-format csv -- python3 profiles/tools/ps_quick.py 8192 1 > $O/sq2.log 2>&1
ls -R $O | head -30

#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 passes over the default bench workload.
#   profiles/tools/profile_bench.sh <outdir under gpurun_out>
# kernel-trace --stats, then separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass; SQ has 8 slots).
# Summarise afterwards with: python profiles/summarize.py gpurun_out/<outdir> r02
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
B="python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pmc --no-paths --no-e2e"
# the stats pass runs the default bench command (python3 bench.py: 20 steps, 3 warm-up, CPU baseline leg) so that
# its average kernel duration is the one bench.py reports from HIP events in the JSON line of the same process
# (kept as bench_under_rocprof.json); the counter passes only need a few launches
rocprofv3 --kernel-trace --stats -d $O/stats/run -o x --output-format csv -- python3 bench.py --no-pmc > $O/stats.log 2>&1
grep '^{"metric"' $O/stats.log > $O/bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch/run -o x --output-format csv -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write/run -o x --output-format csv -- $B > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq/run -o x --output-format csv -- $B > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_sq2/run -o x --output-format csv -- $B > $O/pmc_sq2.log 2>&1
ls -R $O | head -40

# float64 default path (kirch_dquad_kernel) and the secondary paths (Stolt config 2, Gazdag config 5): durations + VALU / wait counters
rocprofv3 --kernel-trace --stats -d $O/stats64/run -o x --output-format csv -- python3 bench.py --dtype f64 --mode exact --steps 5 --no-cpu --no-pmc --no-paths --no-e2e > $O/stats64.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/paths/run -o x --output-format csv -- python3 profiles/tools/paths_quick.py > $O/paths.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/paths_sq/run -o x --output-format csv -- python3 profiles/tools/paths_quick.py > $O/paths_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS -d $O/paths_sq2/run -o x --output-format csv -- python3 profiles/tools/paths_quick.py > $O/paths_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/paths_fetch/run -o x --output-format csv -- python3 profiles/tools/paths_quick.py > $O/paths_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/paths_write/run -o x --output-format csv -- python3 profiles/tools/paths_quick.py > $O/paths_write.log 2>&1
ls -R $O | head -60

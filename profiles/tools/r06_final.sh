#!/bin/bash
# Round 6, final build: the default bench run (plain, then under rocprofv3 --kernel-trace --stats) and the sub-records' kernels.
#   gpurun -- 'bash profiles/tools/r06_final.sh'   ->  gpurun_out/r06/final/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06/final
mkdir -p $O
cd $R
python3 bench.py > $O/bench_plain.log 2>&1
grep '^{"metric"' $O/bench_plain.log > $O/bench_plain_n1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o x --output-format csv -- python3 $R/bench.py --no-pmc > $O/stats.log 2>&1
grep '^{"metric"' $O/stats.log > $O/bench_n1.json
cd $R
rocprofv3 --kernel-trace --stats -d $O/paths -o x --output-format csv -- python3 profiles/tools/paths_quick.py > $O/paths.log 2>&1
ls -R $O | head -30
wc -c $O/bench_plain_n1.json

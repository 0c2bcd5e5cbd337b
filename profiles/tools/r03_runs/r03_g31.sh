cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04b; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tlst -- python3 profiles/tools/stolt_quick.py > $O/tlst_run.txt 2>&1
grep device_ms $O/tlst_run.txt
python3 profiles/tools/timeline.py $O/tlst > $O/timeline_stolt.txt 2>&1
tail -n 45 $O/timeline_stolt.txt
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl64 -- python3 profiles/tools/ps_quick64.py 4096 > $O/tl64_run.txt 2>&1
python3 profiles/tools/timeline.py $O/tl64 > $O/timeline_ps64.txt 2>&1
tail -n 30 $O/timeline_ps64.txt
find $O -name "*.csv" -size +8M -delete

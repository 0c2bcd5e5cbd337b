cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04a; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tlps -- python3 profiles/tools/ps_quick.py 8192 1 > $O/tlps_run.txt 2>&1
tail -n 2 $O/tlps_run.txt
python3 profiles/tools/timeline.py $O/tlps > $O/timeline_ps.txt 2>&1
cat $O/timeline_ps.txt | head -70
find $O/tlps -name "*.csv" -size +8M -delete

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03x; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl -- python3 profiles/tools/e2e_f32.py > $O/tl_run.txt 2>&1
tail -n 5 $O/tl_run.txt
python3 profiles/tools/timeline.py $O/tl > $O/timeline.txt 2>&1
cat $O/timeline.txt | head -80
find $O/tl -name "*.csv" -size +8M -delete

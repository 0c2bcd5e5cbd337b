cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03y; mkdir -p $O; rm -f $O/*.txt
timeout 1800 python -m pytest tests/test_kirchhoff_gpu.py -x -q -k "one_shot" > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
for c3 in 85 80 75 85; do
echo "== cut3=$c3" >> $O/e2e.txt
IMPDAR_KIRCH_ONESHOT_CUT3=$c3 timeout 600 python3 profiles/tools/e2e_phases.py 2>&1 | grep -E "wall" >> $O/e2e.txt
done
cat $O/e2e.txt
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl -- python3 profiles/tools/e2e_f32.py > $O/tl_run.txt 2>&1
python3 profiles/tools/timeline.py $O/tl > $O/timeline.txt 2>&1
cat $O/timeline.txt | head -60
find $O/tl -name "*.csv" -size +8M -delete

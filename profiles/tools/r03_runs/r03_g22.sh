cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x; mkdir -p $O; rm -f $O/e2e.txt
timeout 1800 python -m pytest tests/test_kirchhoff_gpu.py -x -q -k "one_shot or golden or hook or config3 or halo" > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
for sp in 1 2 0 1 2 0; do
echo "== split=$sp" >> $O/e2e.txt
IMPDAR_KIRCH_ONESHOT_SPLIT=$sp timeout 600 python3 profiles/tools/e2e_phases.py 2>&1 | grep -E "wall" >> $O/e2e.txt
done
cat $O/e2e.txt

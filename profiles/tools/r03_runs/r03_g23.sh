cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x; mkdir -p $O; rm -f $O/e2e_t.txt
for sp in 1 2; do
echo "== split=$sp" >> $O/e2e_t.txt
IMPDAR_TIMING=1 IMPDAR_TIMING_SPLIT=1 IMPDAR_KIRCH_ONESHOT_SPLIT=$sp timeout 600 python3 profiles/tools/e2e_phases.py 2>&1 | grep -E "wall|pieces|call" >> $O/e2e_t.txt
done
cat $O/e2e_t.txt

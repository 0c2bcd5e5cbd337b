cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03y; mkdir -p $O; rm -f $O/sweep.txt
for cuts in 10,45,75 10,45,85 10,50,80 6,40,72 10,40,70 12,50,78 8,35,62,84 10,45,70,88 10,55 10,60 10,45,75; do
echo "== cuts=$cuts" >> $O/sweep.txt
E2E_CALLS=9 IMPDAR_KIRCH_ONESHOT_CUTS=$cuts timeout 600 python3 profiles/tools/e2e_f32.py 2>&1 | grep -E "wall" | tail -n 8 | awk '{print $3}' | sort -n | tr '\n' ' ' >> $O/sweep.txt
echo >> $O/sweep.txt
done
cat $O/sweep.txt

cd $GRAFT_REPO_ROOT
O=gpurun_out/r03r; mkdir -p $O
for w in 0 1 2 0 2; do
echo "== warm=$w" >> $O/e2e.txt
IMPDAR_KIRCH_WARM=$w IMPDAR_TIMING=1 timeout 600 python3 profiles/tools/e2e_phases.py 2>&1 | grep -E "cached|wall" | head -8 >> $O/e2e.txt
done
timeout 900 python -m pytest tests/test_kirchhoff_gpu.py -x -q -k "golden or fixture or wrapper or nan" > $O/tests.txt 2>&1; tail -n 2 $O/tests.txt
cat $O/e2e.txt

cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_phaseshift_gpu.py -x -q -k "matrix_core or golden or config5" > $O/tests1.txt 2>&1; echo "pytest rc $?" >> $O/tests1.txt
for rep in 1 2; do
echo "default: $(timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
for v in nomix st8 st16; do
echo "$v: $(IMPDAR_HIP_LIB=$R/build/diag/lib_$v.so timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
done
done
IMPDAR_TIMING=1 timeout 600 python3 profiles/tools/e2e_phases.py > $O/e2e.txt 2>&1
grep -E "passed|failed|Error" $O/tests1.txt | tail -5; cat $O/ps_quick.txt; cat $O/e2e.txt

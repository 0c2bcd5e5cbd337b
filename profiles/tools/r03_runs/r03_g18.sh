cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_phaseshift_gpu.py -x -q > $O/tests1.txt 2>&1; echo "pytest rc $?" >> $O/tests1.txt
for rep in 1 2; do
echo "ch16: $(timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
echo "ch32: $(IMPDAR_HIP_LIB=$R/build/diag/lib_ch32.so timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
done
echo "ch16 4096: $(timeout 600 python3 profiles/tools/ps_quick.py 4096 2 2>&1 | tail -1)" >> $O/ps_quick.txt
echo "ch16 2048: $(timeout 600 python3 profiles/tools/ps_quick.py 2048 2 2>&1 | tail -1)" >> $O/ps_quick.txt
tail -n 3 $O/tests1.txt; cat $O/ps_quick.txt

cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
timeout 2400 python -m pytest tests/test_phaseshift_gpu.py -x -q -s > $O/tests1.txt 2>&1; echo "pytest rc $?" >> $O/tests1.txt
timeout 1800 python -m pytest tests/test_kirchhoff_gpu.py -x -q -k "hook or ties" > $O/tests2.txt 2>&1; echo "pytest rc $?" >> $O/tests2.txt
echo "mfma: $(timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
echo "valu: $(IMPDAR_PS_MFMA=0 timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
echo "mfma 4096: $(timeout 600 python3 profiles/tools/ps_quick.py 4096 2 2>&1 | tail -1)" >> $O/ps_quick.txt
echo "mfma 2048: $(timeout 600 python3 profiles/tools/ps_quick.py 2048 2 2>&1 | tail -1)" >> $O/ps_quick.txt
echo "valu 2048: $(IMPDAR_PS_MFMA=0 timeout 600 python3 profiles/tools/ps_quick.py 2048 2 2>&1 | tail -1)" >> $O/ps_quick.txt
grep -E "passed|failed|rel L2|config 5" $O/tests1.txt | tail -40; tail -n 5 $O/tests2.txt; cat $O/ps_quick.txt

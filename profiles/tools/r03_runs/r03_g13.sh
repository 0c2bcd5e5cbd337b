cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p; mkdir -p $O
timeout 2400 python -m pytest tests/test_phaseshift_gpu.py tests/test_cli_gpu.py tests/test_preproc_gpu.py -x -q > $O/tests1.txt 2>&1; echo "pytest rc $?" >> $O/tests1.txt
timeout 1800 python -m pytest tests/test_kirchhoff_gpu.py -x -q -k "golden or hook or config3 or float32 or nan" > $O/tests2.txt 2>&1; echo "pytest rc $?" >> $O/tests2.txt
echo "mfma: $(timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
IMPDAR_TIMING=1 timeout 600 python3 profiles/tools/e2e_phases.py > $O/e2e.txt 2>&1
tail -n 3 $O/tests1.txt $O/tests2.txt; cat $O/ps_quick.txt; cat $O/e2e.txt

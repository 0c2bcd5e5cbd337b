cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04c; mkdir -p $O; rm -f $O/*.txt
timeout 1800 python -m pytest tests/test_phaseshift_gpu.py -x -q > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt
timeout 600 python3 profiles/tools/ps_quick64.py 8192 >> $O/paths.txt 2>&1
timeout 600 python3 profiles/tools/ps_quick64.py 4096 >> $O/paths.txt 2>&1
IMPDAR_PS_MFMA=0 timeout 600 python3 profiles/tools/ps_quick.py 8192 >> $O/paths.txt 2>&1
cat $O/paths.txt

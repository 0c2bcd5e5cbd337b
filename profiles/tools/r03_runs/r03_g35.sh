cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04f; mkdir -p $O; rm -f $O/*.txt
timeout 1200 python -m pytest tests/test_phaseshift_gpu.py -x -q -k "matrix or mfma or config5 or golden or padded" > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt
timeout 600 python3 profiles/tools/paths_quick.py >> $O/paths.txt 2>&1
timeout 600 python3 profiles/tools/ps_quick.py 8192 >> $O/paths.txt 2>&1
timeout 600 python3 profiles/tools/ps_quick.py 4096 >> $O/paths.txt 2>&1
cat $O/paths.txt

cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04i; mkdir -p $O; rm -f $O/*.txt
IMPDAR_PS_MFMA_SPEC=1 timeout 1200 python -m pytest tests/test_phaseshift_gpu.py -x -q -k "matrix or mfma or config5 or golden or padded" > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt; grep -E "^E " $O/tests.txt | head -5
for sp in 0 1 0 1; do
echo "== IMPDAR_PS_MFMA_SPEC=$sp" >> $O/paths.txt
IMPDAR_PS_MFMA_SPEC=$sp timeout 600 python3 profiles/tools/ps_quick.py 8192 >> $O/paths.txt 2>&1
IMPDAR_PS_MFMA_SPEC=$sp timeout 600 python3 profiles/tools/ps_quick.py 4096 >> $O/paths.txt 2>&1
done
cat $O/paths.txt

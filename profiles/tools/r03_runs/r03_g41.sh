cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04k; mkdir -p $O; rm -f $O/*.txt
timeout 1200 python -m pytest tests/test_phaseshift_gpu.py -x -q -k "matrix or mfma or config5 or golden or padded" > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt; grep -E "^E " $O/tests.txt | head -5
for v in ps_fuse0 ps_fuse1 ps_fuse0 ps_fuse1; do
echo "== $v"
IMPDAR_HIP_LIB=$PWD/build/diag/lib_$v.so python3 profiles/tools/ps_quick.py 8192 | tail -1
done
IMPDAR_HIP_LIB=$PWD/build/diag/lib_ps_fuse0.so python3 profiles/tools/ps_quick.py 4096 | tail -1
IMPDAR_HIP_LIB=$PWD/build/diag/lib_ps_fuse1.so python3 profiles/tools/ps_quick.py 4096 | tail -1

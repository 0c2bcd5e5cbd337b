cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04a; mkdir -p $O; rm -f $O/*.txt
build/probe/fft_strided_probe 8192 > $O/fft_probe.txt 2>&1
timeout 1800 python -m pytest tests/test_phaseshift_gpu.py tests/test_phaseshift_sharded_gpu.py -x -q > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt
for f in rows strided rows; do
echo "== IMPDAR_PS_FFT=$f" >> $O/paths.txt
IMPDAR_PS_FFT=$f timeout 600 python3 profiles/tools/paths_quick.py >> $O/paths.txt 2>&1
IMPDAR_PS_FFT=$f timeout 600 python3 profiles/tools/ps_quick.py 8192 >> $O/paths.txt 2>&1
IMPDAR_PS_FFT=$f timeout 600 python3 profiles/tools/ps_quick64.py 8192 >> $O/paths.txt 2>&1
done
cat $O/paths.txt

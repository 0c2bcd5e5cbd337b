cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_phaseshift_gpu.py -x -q -s > $O/tests1.txt 2>&1; echo "pytest rc $?" >> $O/tests1.txt
echo "mfma: $(timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
echo "mfma 4096: $(timeout 600 python3 profiles/tools/ps_quick.py 4096 2 2>&1 | tail -1)" >> $O/ps_quick.txt
echo "mfma 2048: $(timeout 600 python3 profiles/tools/ps_quick.py 2048 2 2>&1 | tail -1)" >> $O/ps_quick.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O/ps_stats/run -o x --output-format csv -- python3 $R/profiles/tools/ps_quick.py 8192 1 > $R/$O/ps_stats.log 2>&1
cp $(find $R/$O/ps_stats -name '*kernel_stats.csv' | head -1) $R/$O/ps_kernel_stats.csv
cd $R
grep -E "passed|failed|Error|rel L2|config 5" $O/tests1.txt | tail -24; cat $O/ps_quick.txt; head -8 $O/ps_kernel_stats.csv | cut -c1-150

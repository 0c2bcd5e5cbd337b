cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout 1500 python -m pytest tests/test_phaseshift_gpu.py tests/test_comm_gpu.py -x -q > $O/tests1.txt 2>&1; echo "pytest rc $?" >> $O/tests1.txt
timeout 1800 python -m pytest tests/test_kirchhoff_gpu.py -x -q -k "hook or ties or config3 or golden" > $O/tests2.txt 2>&1; echo "pytest rc $?" >> $O/tests2.txt
echo "default: $(timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
echo "vz32w4: $(IMPDAR_HIP_LIB=$PWD/build/diag/lib_vz32w4.so timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
echo "default 4096: $(timeout 600 python3 profiles/tools/ps_quick.py 4096 2 2>&1 | tail -1)" >> $O/ps_quick.txt
echo "f64: $(timeout 600 python3 profiles/tools/ps_quick64.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
timeout 1200 python3 profiles/tools/exchange_overlap.py c4r8 c3r8 c3r4 c3r2 --reserve 0,8,16,32,64 > $O/exchange_overlap.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for R in 0 16; do
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_r$R/run -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/profiles/tools/exchange_overlap.py c4r8 c3r8 --reserve $R --trace > $GRAFT_REPO_ROOT/$O/trace_r$R.log 2>&1
f=$(find $GRAFT_REPO_ROOT/$O/trace_r$R -name '*kernel_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/profiles/tools/trace_overlap.py $f 60 > $GRAFT_REPO_ROOT/$O/trace_overlap_r$R.txt 2>&1
done
cd $GRAFT_REPO_ROOT
tail -3 $O/tests1.txt $O/tests2.txt; cat $O/ps_quick.txt; cat $O/exchange_overlap.txt; head -30 $O/trace_overlap_r0.txt; head -30 $O/trace_overlap_r16.txt

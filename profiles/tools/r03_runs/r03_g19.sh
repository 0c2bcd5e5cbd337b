cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
timeout 1500 python3 tests/tools/fuzz_kirchhoff.py 1200 41 > $O/fuzz_k.txt 2>&1; echo "rc $?" >> $O/fuzz_k.txt
timeout 2400 python3 tests/tools/fuzz_phaseshift.py 500 42 > $O/fuzz_ps.txt 2>&1; echo "rc $?" >> $O/fuzz_ps.txt
timeout 900 python3 tests/tools/fuzz_stolt.py 150 43 > $O/fuzz_s.txt 2>&1; echo "rc $?" >> $O/fuzz_s.txt
grep "^cases" $O/fuzz_k.txt $O/fuzz_ps.txt $O/fuzz_s.txt

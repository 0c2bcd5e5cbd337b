cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
timeout 2400 python3 tests/tools/fuzz_phaseshift.py 400 31 > $O/fuzz_ps.txt 2>&1; echo "rc $?" >> $O/fuzz_ps.txt
timeout 1500 python3 tests/tools/fuzz_kirchhoff.py 400 32 > $O/fuzz_k.txt 2>&1; echo "rc $?" >> $O/fuzz_k.txt
timeout 900 python3 tests/tools/fuzz_stolt.py 100 33 > $O/fuzz_s.txt 2>&1; echo "rc $?" >> $O/fuzz_s.txt
tail -n 4 $O/fuzz_ps.txt $O/fuzz_k.txt $O/fuzz_s.txt

cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O; rm -f $O/*.txt
timeout 1200 python3 tests/tools/fuzz_kirchhoff.py 1500 61 > $O/fuzz_k.txt 2>&1; echo "rc $?" >> $O/fuzz_k.txt
timeout 1500 python3 tests/tools/fuzz_phaseshift.py 600 62 > $O/fuzz_ps.txt 2>&1; echo "rc $?" >> $O/fuzz_ps.txt
timeout 600 python3 tests/tools/fuzz_stolt.py 200 63 > $O/fuzz_s.txt 2>&1; echo "rc $?" >> $O/fuzz_s.txt
timeout 300 python3 tests/tools/fuzz_ps_sharded.py 600 64 > $O/fuzz_sh.txt 2>&1; echo "rc $?" >> $O/fuzz_sh.txt
timeout 600 python3 tests/tools/fuzz_oneshot_pieces.py 300 65 > $O/fuzz_os.txt 2>&1; echo "rc $?" >> $O/fuzz_os.txt
grep "^cases" $O/*.txt

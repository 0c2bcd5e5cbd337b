cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
timeout 900 python3 -m pytest tests/test_phaseshift_sharded_gpu.py -q > $O/sharded_tests.txt 2>&1; echo "rc $?" >> $O/sharded_tests.txt
timeout 600 python3 profiles/tools/ps_sharded_emulate.py 8192 > $O/ps_sharded_emulate.txt 2>&1; echo "rc $?" >> $O/ps_sharded_emulate.txt
tail -n 15 $O/sharded_tests.txt; cat $O/ps_sharded_emulate.txt

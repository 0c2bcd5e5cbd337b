cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
timeout 900 python3 -m pytest tests/test_phaseshift_sharded_gpu.py -x -q > $O/sharded_tests.txt 2>&1; echo "rc $?" >> $O/sharded_tests.txt
timeout 900 python3 -m pytest tests/test_phaseshift_gpu.py tests/test_comm_gpu.py -x -q > $O/ps_tests.txt 2>&1; echo "rc $?" >> $O/ps_tests.txt
timeout 600 python3 profiles/tools/ps_sharded_emulate.py 8192 > $O/ps_sharded_emulate.txt 2>&1; echo "rc $?" >> $O/ps_sharded_emulate.txt
timeout 120 build/probe/h2d_2d_probe > $O/h2d_probe.txt 2>&1; echo "rc $?" >> $O/h2d_probe.txt
tail -5 $O/sharded_tests.txt $O/ps_tests.txt; cat $O/ps_sharded_emulate.txt $O/h2d_probe.txt

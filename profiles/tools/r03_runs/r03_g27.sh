cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03z; mkdir -p $O; rm -f $O/*.txt
timeout 2400 python -m pytest tests/test_kirchhoff_gpu.py tests/test_comm_gpu.py tests/test_cli_gpu.py -x -q > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
timeout 600 python3 profiles/tools/e2e_phases.py 2>&1 | grep wall > $O/e2e.txt; cat $O/e2e.txt

cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04_final4
O=gpurun_out/r04_final4
timeout 2200 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; grep -E "passed|failed" $O/gpu_tests.txt | tail -1
(timeout 1200 python tests/tools/fuzz_phaseshift.py 400 101 2>&1 | tail -6) > $O/fuzz_ps.txt; tail -1 $O/fuzz_ps.txt
(timeout 900 python tests/tools/fuzz_ps_sharded.py 150 102 2>&1 | tail -4) > $O/fuzz_pss.txt; tail -1 $O/fuzz_pss.txt
(timeout 600 python tests/tools/fuzz_kirchhoff.py 200 103 2>&1 | tail -4) > $O/fuzz_k.txt; tail -1 $O/fuzz_k.txt
(timeout 600 python tests/tools/fuzz_oneshot_pieces.py 40 104 2>&1 | tail -2) > $O/fuzz_one.txt; tail -1 $O/fuzz_one.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1

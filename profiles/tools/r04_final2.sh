cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04_final2
O=gpurun_out/r04_final2
timeout 900 python -m pytest tests/test_phaseshift_gpu.py -q -m gpu -k "every_step or changing_in_most or hands_over" 2>&1 | tail -2
timeout 300 python profiles/tools/ps_smooth.py 4096 2>&1 | tail -1
timeout 300 python profiles/tools/ps_smooth.py 8192 2>&1 | tail -1
timeout 900 python -m pytest tests/test_kirchhoff_gpu.py -q -m gpu -k "general_geometry or jittered or fast_mode" 2>&1 | tail -2
(timeout 600 python profiles/tools/gen_quick.py 2>&1 | tail -45) > $O/gen_quick.txt; grep -c " ok" $O/gen_quick.txt; grep MISS $O/gen_quick.txt | head -3; tail -2 $O/gen_quick.txt
(timeout 1200 python tests/tools/fuzz_phaseshift.py 300 82 2>&1 | tail -8) > $O/fuzz_ps.txt; tail -7 $O/fuzz_ps.txt | cut -c1-200

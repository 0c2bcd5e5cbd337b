cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04_final
O=gpurun_out/r04_final
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.log
echo "bench rc=$?"
(timeout 900 python tests/tools/fuzz_kirchhoff.py 400 71 2>&1 | tail -8) > $O/fuzz_k.txt
(timeout 900 python tests/tools/fuzz_phaseshift.py 300 72 2>&1 | tail -8) > $O/fuzz_ps.txt
(timeout 600 python tests/tools/fuzz_stolt.py 100 73 2>&1 | tail -4) > $O/fuzz_s.txt
(timeout 600 python tests/tools/fuzz_oneshot_pieces.py 40 74 2>&1 | tail -3) > $O/fuzz_one.txt
(timeout 600 python profiles/tools/gen_quick.py 2>&1 | tail -45) > $O/gen_quick.txt
tail -2 $O/fuzz_k.txt $O/fuzz_ps.txt $O/fuzz_s.txt $O/fuzz_one.txt | cut -c1-300
cut -c1-300 $O/bench_default.json

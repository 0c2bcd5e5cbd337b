cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
IMPDAR_PS_MFMA_ONE_WG=1 IMPDAR_PS_STAMPS=1 timeout 600 python3 profiles/tools/ps_quick.py 8192 1 > $O/stamps_one.txt 2>&1
IMPDAR_PS_STAMPS=1 timeout 600 python3 profiles/tools/ps_quick.py 8192 1 > $O/stamps_two.txt 2>&1
grep "stamps round  [67]" $O/stamps_one.txt | head -8; tail -1 $O/stamps_one.txt | cut -c1-300
grep "stamps round  [67]" $O/stamps_two.txt | head -8; tail -1 $O/stamps_two.txt | cut -c1-300
IMPDAR_BENCH_FORCE_DIST=1 timeout 900 python3 bench.py --no-cpu --no-paths --no-pmc --no-e2e --steps 5 > $O/bench_dist.json 2> $O/bench_dist.err; echo "force-dist bench rc $?"; cut -c1-600 $O/bench_dist.json

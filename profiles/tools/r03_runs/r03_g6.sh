cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/tests_all.txt 2>&1; echo "pytest rc $?" >> $O/tests_all.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
echo "mfma: $(timeout 600 python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)" >> $O/ps_quick.txt
tail -n 4 $O/tests_all.txt; tail -n 3 $O/smoke.txt; tail -n 12 $O/bench.err; cat $O/bench.json | cut -c1-6000; cat $O/ps_quick.txt

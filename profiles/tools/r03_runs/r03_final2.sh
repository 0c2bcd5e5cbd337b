cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/final2_r03; mkdir -p $O; rm -rf $O/*
timeout 3000 python -m pytest tests/ -q -m gpu > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; python3 -c "
import json
r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print({k:r[k] for k in ('value','ms_per_step')}, r['roofline']['frac'], r['end_to_end']['wall_ms'], {k:(v.get('device_ms'),v.get('kernel_ms')) for k,v in r['paths'].items()})
"
bash profiles/tools/profile_bench.sh prof_r03 > $O/profile.log 2>&1; tail -n 2 $O/profile.log
timeout 600 python3 tests/tools/fuzz_kirchhoff.py 300 51 > $O/fuzz_k.txt 2>&1; tail -n 1 $O/fuzz_k.txt
timeout 900 python3 tests/tools/fuzz_phaseshift.py 150 52 > $O/fuzz_ps.txt 2>&1; tail -n 1 $O/fuzz_ps.txt
timeout 400 python3 tests/tools/fuzz_stolt.py 60 53 > $O/fuzz_s.txt 2>&1; tail -n 1 $O/fuzz_s.txt
timeout 400 python3 tests/tools/fuzz_ps_sharded.py 150 54 > $O/fuzz_sh.txt 2>&1; tail -n 1 $O/fuzz_sh.txt

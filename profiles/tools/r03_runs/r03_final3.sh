cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/final3_r03; mkdir -p $O; rm -rf $O/*
timeout 3000 python -m pytest tests/ -q -m gpu > $O/tests.txt 2>&1; grep -E "passed|failed" $O/tests.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; python3 -c "
import json
r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print({k:r[k] for k in ('value','ms_per_step')}, r['roofline']['frac'], r['end_to_end']['wall_ms'], r['end_to_end']['wall_ms_fresh_array'], {k:(v.get('device_ms'),v.get('kernel_ms')) for k,v in r['paths'].items()})
"
bash profiles/tools/profile_bench.sh prof_r03 > $O/profile.log 2>&1; tail -n 2 $O/profile.log

#!/bin/bash
# same-box A/B of environment knobs:  env_run.sh "IMPDAR_KIRCH_NH=1" "IMPDAR_KIRCH_NH=2" ...  (each runs twice)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for e in "$@"; do
  for rep in 1 2; do
    env $e python bench.py --no-cpu --no-pmc --no-paths --no-e2e --steps 10 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('$e', 'ms/step %.3f kernel %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
  done
done

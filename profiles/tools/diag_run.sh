#!/bin/bash
# usage: diag_run.sh name1 name2 ... (libs under build/diag/lib_<name>.so; "base" = product lib)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for n in "$@"; do
  if [ "$n" = base ]; then L=impdar_amd/csrc/libimpdar_hip.so; else L=build/diag/lib_$n.so; fi
  for rep in 1 2; do
    IMPDAR_HIP_LIB=$PWD/$L python bench.py --no-cpu --no-pmc --no-paths --no-e2e --steps 10 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('$n', 'ms/step %.3f kernel %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
  done
done

#!/bin/bash
# same-box A/B of phase-shift library variants (build/diag/lib_<name>.so, SRC=phaseshift variant_build.sh): device ms at config 5
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  echo "default: $(python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)"
  for n in "$@"; do
    echo "$n: $(IMPDAR_HIP_LIB=$R/build/diag/lib_$n.so python3 profiles/tools/ps_quick.py 8192 2 2>&1 | tail -1)"
  done
done

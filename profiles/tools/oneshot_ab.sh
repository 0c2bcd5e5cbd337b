#!/bin/bash
# one-shot Kirchhoff at config 3 (float32 host array in, float64 out, vel 1.69e8 as bench.py): call_ms / kernel span
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" IMPDAR_METRICS=1 E2E_CALLS=6 timeout 300 python profiles/tools/e2e_f32.py 2>&1 | grep -o '"kernel_ms": [0-9.]*\|"call_ms": [0-9.]*\|wall [0-9.]* ms' | paste - - - | tail -4; }
run A=1
run IMPDAR_KIRCH_ONESHOT_SPLIT=1
run IMPDAR_KIRCH_ONESHOT_SPLIT=0
run IMPDAR_KIRCH_RESERVE=16
run IMPDAR_KIRCH_RESERVE=8
run A=1

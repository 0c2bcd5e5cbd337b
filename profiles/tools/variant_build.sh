#!/bin/bash
# Build a variant of the library with extra compile flags for kirchhoff.hip (same-box A/B runs, diagnostics):
#   profiles/tools/variant_build.sh <name> "<flags>" [<name> "<flags>" ...]   ->  build/diag/lib_<name>.so
# SRC=phaseshift (stolt, preproc ...) rebuilds that source instead of kirchhoff.hip.
# Run one with IMPDAR_HIP_LIB=$PWD/build/diag/lib_<name>.so python bench.py --no-cpu  (profiles/tools/diag_run.sh)
R=$(cd "$(dirname "$0")/../.." && pwd)
SRC=${SRC:-kirchhoff}
OBJS=""
for o in api comm kirchhoff kirch_gen stolt phaseshift preproc; do [ $o != $SRC ] && OBJS="$OBJS $R/impdar_amd/csrc/$o.o"; done
mkdir -p $R/build/diag
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function \
      $f -c $R/impdar_amd/csrc/$SRC.hip -o $R/build/diag/k_$n.o 2>&1 | grep -E "error" -A3
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $R/build/diag/k_$n.o -o $R/build/diag/lib_$n.so -L/opt/rocm/lib -lrocfft -lrccl -Wl,-rpath,/opt/rocm/lib ) &
done
wait
ls -la $R/build/diag/*.so | awk '{print $5, $9}'

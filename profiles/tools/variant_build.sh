#!/bin/bash
# Build a variant of the library with extra compile flags for kirchhoff.hip (same-box A/B runs, diagnostics):
#   profiles/tools/variant_build.sh <name> "<flags>" [<name> "<flags>" ...]   ->  build/diag/lib_<name>.so
# Run one with IMPDAR_HIP_LIB=$PWD/build/diag/lib_<name>.so python bench.py --no-cpu  (profiles/tools/diag_run.sh)
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/build/diag
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function \
      $f -c $R/impdar_amd/csrc/kirchhoff.hip -o $R/build/diag/k_$n.o 2>&1 | grep -E "error" -A3
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $R/impdar_amd/csrc/api.o $R/impdar_amd/csrc/comm.o $R/build/diag/k_$n.o \
      $R/impdar_amd/csrc/stolt.o $R/impdar_amd/csrc/phaseshift.o $R/impdar_amd/csrc/preproc.o -o $R/build/diag/lib_$n.so -L/opt/rocm/lib -lrocfft -lrccl -Wl,-rpath,/opt/rocm/lib ) &
done
wait
ls -la $R/build/diag/*.so | awk '{print $5, $9}'

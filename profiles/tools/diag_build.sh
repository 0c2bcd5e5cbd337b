#!/bin/bash
# Build diagnostic variants of the library (timing ablations of kirch_quad_kernel; results are NOT valid):
#   profiles/tools/diag_build.sh "NOLDS" "NOFMA -DKQ_DIAG_NOSTAGE" ...   ->  build/diag/lib_<name>.so
# Run one with IMPDAR_HIP_LIB=$PWD/build/diag/lib_<name>.so python bench.py --no-cpu
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/build/diag
for v in "$@"; do
  n=$(echo $v | tr -d ' ' | sed 's/-DKQ_DIAG_/_/g; s/-D//g')
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function \
      -DKQ_DIAG_$v -c $R/impdar_amd/csrc/kirchhoff.hip -o $R/build/diag/k_$n.o 2>&1 | grep -E "error" -A3
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $R/impdar_amd/csrc/api.o $R/impdar_amd/csrc/comm.o $R/build/diag/k_$n.o \
      $R/impdar_amd/csrc/stolt.o $R/impdar_amd/csrc/phaseshift.o $R/impdar_amd/csrc/preproc.o -o $R/build/diag/lib_$n.so -L/opt/rocm/lib -lrocfft -lrccl -Wl,-rpath,/opt/rocm/lib ) &
done
wait
ls $R/build/diag/*.so

# Band-pass ablations (csrc/preproc.hip, -DFF_DIAG_NOSTORE / NOLOAD / NODPP): set FF="NOSTORE NOLOAD ..." in the environment
for v in $FF; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function \
      -DFF_DIAG_$v -c $R/impdar_amd/csrc/preproc.hip -o $R/build/diag/p_$v.o 2>&1 | grep -E "error" -A3
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $R/impdar_amd/csrc/api.o $R/impdar_amd/csrc/comm.o $R/impdar_amd/csrc/kirchhoff.o \
      $R/impdar_amd/csrc/stolt.o $R/impdar_amd/csrc/phaseshift.o $R/build/diag/p_$v.o -o $R/build/diag/lib_ff_$v.so -L/opt/rocm/lib -lrocfft -lrccl -Wl,-rpath,/opt/rocm/lib ) &
done
wait

#!/bin/bash
# GPU box: A/B builds of phaseshift.hip with -D switches (timing experiments).
#   profiles/tools/ps_variant.sh "<python script and args>" VARIANT [VARIANT ...]    VARIANT = FULL or -D names joined by '+'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/impdar_amd/csrc
CMD=$1
shift
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function"
OBJS="api.o comm.o kirchhoff.o kirch_gen.o stolt.o preproc.o"
for v in "$@"; do
  D=$(echo $v | sed 's/+/ -D/g')
  /opt/rocm/bin/hipcc $FLAGS -D$D -c phaseshift.hip -o /tmp/ps_$v.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/ps_$v.o -o /tmp/libimpdar_$v.so -L/opt/rocm/lib -lrocfft -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
  echo "== $v"
  (cd $R && IMPDAR_HIP_LIB=/tmp/libimpdar_$v.so python3 $CMD)
done

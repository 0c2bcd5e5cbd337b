#!/bin/bash
# GPU box: A/B builds of ONE source of the library with -D switches (timing experiments).
#   profiles/tools/variant.sh <source.hip> "<python script and args>" VARIANT [VARIANT ...]    VARIANT = FULL or -D names joined by '+'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/impdar_amd/csrc
SRC=$1
CMD=$2
shift 2
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function"
OBJS=""
for o in api comm kirchhoff kirch_gen stolt phaseshift preproc; do [ "$o.hip" != "$SRC" ] && OBJS="$OBJS $o.o"; done
for v in "$@"; do
  D=$(echo $v | sed 's/+/ -D/g')
  /opt/rocm/bin/hipcc $FLAGS -D$D -c $SRC -o /tmp/v_$v.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/v_$v.o -o /tmp/libimpdar_$v.so -L/opt/rocm/lib -lrocfft -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
  echo "== $v"
  (cd $R && IMPDAR_HIP_LIB=/tmp/libimpdar_$v.so python3 $CMD)
done

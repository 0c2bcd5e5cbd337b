#!/bin/bash
# GPU box: timing-only builds of ps_runs_kernel (results wrong by construction): where its time goes.
#   profiles/tools/ps_runs_abl.sh <rows> [variant ...]   variants are sets of -D switches joined by '+', e.g. PR_ABL_NOSETUP+PR_ABL_NOMFMA
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/impdar_amd/csrc
ROWS=${1:-41}
shift
VARS=${@:-FULL PR_ABL_NOSETUP PR_ABL_NOITEMS PR_ABL_NOMFMA}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function"
OBJS="api.o comm.o kirchhoff.o kirch_gen.o stolt.o preproc.o"
for v in $VARS; do
  D=$(echo $v | sed 's/+/ -D/g')
  /opt/rocm/bin/hipcc $FLAGS -D$D -c phaseshift.hip -o /tmp/ps_$v.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/ps_$v.o -o /tmp/libimpdar_$v.so -L/opt/rocm/lib -lrocfft -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
  echo "== $v"
  (cd $R && IMPDAR_HIP_LIB=/tmp/libimpdar_$v.so python3 profiles/tools/ps_table_quick.py 8192 $ROWS 2)
done

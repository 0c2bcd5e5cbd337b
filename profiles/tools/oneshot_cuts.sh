R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/impdar_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function"
OBJS="api.o comm.o kirch_gen.o stolt.o phaseshift.o preproc.o"
i=0
for v in "{10,40,70}" "{10,35,60,85}" "{8,30,55,80}" "{10,45,80}" "{12,45,78}" "{10,40,65,88}"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc $FLAGS "-DKOS_PCT=$v" -c kirchhoff.hip -o /tmp/k_$i.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/k_$i.o -o /tmp/libimpdar_k$i.so -L/opt/rocm/lib -lrocfft -lrccl -Wl,-rpath,/opt/rocm/lib || exit 1
  echo "== $v"
  (cd $R && E2E_CALLS=7 IMPDAR_HIP_LIB=/tmp/libimpdar_k$i.so python3 profiles/tools/e2e_f32.py | tr '\n' ' '; echo)
done

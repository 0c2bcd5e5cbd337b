cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { # lib env...
  L=$1; shift
  for rep in 1 2; do
    env IMPDAR_HIP_LIB=$PWD/$L "$@" python bench.py --no-cpu --no-pmc --no-paths --no-e2e --steps 10 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('$L $*', 'ms/step %.3f kernel %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
  done
}
run impdar_amd/csrc/libimpdar_hip.so IMPDAR_KIRCH_XB=32 IMPDAR_KIRCH_NH=2
run build/diag/lib_ahead1.so IMPDAR_KIRCH_XB=32 IMPDAR_KIRCH_NH=2
run impdar_amd/csrc/libimpdar_hip.so IMPDAR_KIRCH_NH=1
run build/diag/lib_ahead1.so IMPDAR_KIRCH_NH=1
run impdar_amd/csrc/libimpdar_hip.so IMPDAR_KIRCH_XB=32 IMPDAR_KIRCH_NH=2

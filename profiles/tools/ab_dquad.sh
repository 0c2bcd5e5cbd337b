cd ${GRAFT_REPO_ROOT:-/root/repo}
for e in "IMPDAR_KIRCH_NHD=1" "IMPDAR_KIRCH_NHD=2" "IMPDAR_KIRCH_NHD=2 IMPDAR_KIRCH_XBD=20" "IMPDAR_KIRCH_NHD=1 IMPDAR_KIRCH_XBD=16" "IMPDAR_KIRCH_NHD=1"; do
  for rep in 1 2; do
    env $e python bench.py --dtype f64 --mode exact --no-cpu --no-pmc --no-paths --no-e2e --steps 6 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('$e', 'ms/step %.3f kernel %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
  done
done

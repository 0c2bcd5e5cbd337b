cd ${GRAFT_REPO_ROOT:-/root/repo}
for e in "IMPDAR_KIRCH_XB=24 IMPDAR_KIRCH_NH=1" "IMPDAR_KIRCH_XB=40 IMPDAR_KIRCH_NH=1" "IMPDAR_KIRCH_XB=32 IMPDAR_KIRCH_NH=2"; do
  echo "== $e"; env $e python profiles/tools/rank_steps.py 2 4 8 2>&1 | grep -v "per-trace cost"
done

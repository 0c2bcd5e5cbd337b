R=${GRAFT_REPO_ROOT:-/root/repo}
for L in old new old new; do
  if [ $L = old ]; then export IMPDAR_HIP_LIB=$R/impdar_amd/csrc/libimpdar_hip_old.so; else unset IMPDAR_HIP_LIB; fi
  echo "== $L"; python3 profiles/tools/r06_e2e_trace.py 2>&1 | grep -E "^---- (phsh|wall)" | tr '\n' ' '; echo
done
unset IMPDAR_HIP_LIB
python3 profiles/tools/r06_e2e_trace.py 2>&1 | awk '/phsh call 3/,/wall/' | head -60

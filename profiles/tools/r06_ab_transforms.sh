# same-box A/B of two builds of the library (IMPDAR_HIP_LIB = the other build): phase shift 8192^2 float32 / float64, Stolt 4096^2,
# then the kernels of one run of each under rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-/root/repo}
OLD=$R/impdar_amd/csrc/libimpdar_hip_old.so
mkdir -p gpurun_out/r06
for rep in 1 2; do
for L in old new; do
  if [ $L = old ]; then export IMPDAR_HIP_LIB=$OLD; else unset IMPDAR_HIP_LIB; fi
  echo "== $L"; timeout 300 python3 profiles/tools/ps_quick.py 8192 5 | cut -c1-400
  timeout 300 python3 profiles/tools/ps_quick64.py 8192 3 | cut -c1-400
  timeout 300 python3 profiles/tools/stolt_quick.py | tail -1
done
done
cd /tmp && export TMPDIR=/tmp
for L in old new; do
  if [ $L = old ]; then export IMPDAR_HIP_LIB=$OLD; else unset IMPDAR_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06/abt_$L -o x --output-format csv -- python3 $R/profiles/tools/ps_quick.py 8192 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06/abt64_$L -o x --output-format csv -- python3 $R/profiles/tools/ps_quick64.py 8192 3 > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv,glob
for L in ('old','new'):
  for d in ('abt','abt64'):
    for f in glob.glob("gpurun_out/r06/%s_%s/**/*kernel_stats.csv"%(d,L), recursive=True):
        for r in list(csv.DictReader(open(f)))[:11]:
            print(L, d, "%-72s %4s %10.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])/1e3))
PY

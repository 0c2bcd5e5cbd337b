R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for e in own; do
IMPDAR_PS_FFT=$e rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_psown2 -o x --output-format csv -- python3 $R/profiles/tools/ps_quick.py 8192 2 > /dev/null 2>&1
done
cd $R
python3 - <<EOF
import csv,glob
for f in glob.glob("gpurun_out/r05_psown2/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        if "own_fft" in r["Name"]: print("%-72s %4s %10.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])/1e3))
EOF
IMPDAR_PS_FFT=own python3 profiles/tools/ps_quick.py 8192 3 | cut -c1-200
python3 profiles/tools/stolt_quick.py | tail -3
